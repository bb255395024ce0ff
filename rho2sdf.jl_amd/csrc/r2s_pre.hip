// Pre-stage of the hot path on gfx950:
//   calculate_mesh_volume      (src/MeshGrid/MeshVolume.jl:4-72)         HEX8
//   DenseInNodes               (src/MeshGrid/NodalDensities.jl:89-218)
//   find_threshold_for_volume  (src/MeshGrid/Isocontour_volume.jl:1-154) HEX8 (the reference
//                                hard-codes 8 nodes: TET4 needs an explicit threshold)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "r2s_common.hpp"
#include "r2s_device_math.hpp"
#include "r2s_internal.hpp"

using namespace r2s;

extern "C" void r2s_internal_gauss_legendre(int n, double* x, double* w);

struct GaussTab {
    double gp[16];
    double gw[16];
};

__device__ __forceinline__ double det3(const double J[3][3])
{
    return J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
           J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
}

// one Gauss point of an element: w_i w_j w_k |det J| (0 if the interpolated density is below thr)
__device__ __forceinline__ double quad_point(const double Xe[8][3], const double re[8], double gx, double gy, double gz,
                                             double wgt, bool check, double thr)
{
    double xi[3] = {gx, gy, gz}, N[8], dN[8][3], J[3][3];
    hex8_shape_d(xi, N, dN);
    if (check) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < 8; ++a) v += N[a] * re[a];
        if (v < thr) return 0.0;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 8; ++a) s += Xe[a][r] * dN[a][c];
            J[r][c] = s;
        }
    return wgt * fabs(det3(J));
}

// One wavefront per element.  mode 0: full 3^3 volume of every element (MeshVolume.jl:45-72);
// mode 1: volume of {rho >= thr} (Isocontour_volume.jl:22-71): skip / 3^3 / 15^3 with point test.
// The lanes split the Gauss points; lane partial sums are combined by a fixed butterfly.
__global__ void __launch_bounds__(256) elem_volume_kernel(const double* __restrict__ X, const int64_t* __restrict__ IEN,
                                                         const double* __restrict__ rho_n, int64_t nel, int mode,
                                                         double thr, GaussTab g3, GaussTab g15, double* __restrict__ vol)
{
    const int64_t e = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (e >= nel) return;
    const int lane = threadIdx.x & 63;
    double Xe[8][3], re[8], mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int64_t n = IEN[e * 8 + a] - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) Xe[a][i] = X[3 * n + i];
        re[a] = mode ? rho_n[n] : 0.0;
        mn = fmin(mn, re[a]);
        mx = fmax(mx, re[a]);
    }
    double acc = 0.0;
    if (mode == 0 || mn >= thr) {
        if (lane < 27) {
            const int i = lane % 3, j = (lane / 3) % 3, k = lane / 9;
            acc = quad_point(Xe, re, g3.gp[i], g3.gp[j], g3.gp[k], g3.gw[i] * g3.gw[j] * g3.gw[k], false, thr);
        }
    } else if (!(mx < thr)) {
        for (int p = lane; p < 3375; p += 64) {
            const int i = p % 15, j = (p / 15) % 15, k = p / 225;
            acc += quad_point(Xe, re, g15.gp[i], g15.gp[j], g15.gp[k], g15.gw[i] * g15.gw[j] * g15.gw[k], true, thr);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) vol[e] = acc;
}

// calculate_element_volume, TET4 (MeshVolume.jl:75-117): cube Gauss points collapsed onto the unit
// tetrahedron.  NOTE: the reference's `jacobian_transform = (1-xi)^2 (1-xi-eta)/8` (:108) carries one
// factor (1-xi) too many, so every TET4 volume comes out 25 % low (V_frac is unaffected); restated as is.
__global__ void tet_volume_kernel(const double* __restrict__ X, const int64_t* __restrict__ IEN, int64_t nel,
                                  GaussTab g3, double* __restrict__ vol)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    double xe[4][3], J[3][3];
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 3; ++i) xe[a][i] = X[3 * (IEN[e * 4 + a] - 1) + i];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int a = 0; a < 4; ++a) {
                const double dn = (a == c) ? 1.0 : ((a == 3) ? -1.0 : 0.0);
                s += xe[a][r] * dn;
            }
            J[r][c] = s;
        }
    const double adet = fabs(det3(J));
    double v = 0.0;
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                const double xi = (g3.gp[i] + 1.0) / 2.0;
                const double eta = (g3.gp[j] + 1.0) / 2.0 * (1.0 - xi);
                const double zeta = (g3.gp[k] + 1.0) / 2.0 * (1.0 - xi - eta);
                if (xi < 0 || eta < 0 || zeta < 0 || xi + eta + zeta > 1.0) continue;
                const double jt = (1.0 - xi) * (1.0 - xi) * (1.0 - xi - eta) / 8.0;
                v += g3.gw[i] * g3.gw[j] * g3.gw[k] * adet * jt;
            }
    vol[e] = v;
}

// TET4 iso-volume (the reference has none: calculate_isocontour_volume hard-codes 8 nodes, Isocontour_volume.jl:27-38;
// SURVEY 8(f)2).  Built from the reference's own pieces so that volume(thr = 0) equals calculate_mesh_volume's
// V_domain for TET4 (which the target V_domain*V_frac is expressed in): the classification of :40-52 (skip /
// whole element / cut element) with the collapsed-cube rule of MeshVolume.jl:75-117 - 3^3 points for whole
// elements, 15^3 points with the point test N(xi).rho_e >= thr (:62-64) for cut ones, the same
// `jacobian_transform` (and with it the same 25 % deficit) in both.  One wavefront per element, lanes split the
// points, fixed butterfly.
__global__ void __launch_bounds__(256) tet_iso_volume_kernel(const double* __restrict__ X, const int64_t* __restrict__ IEN,
                                                            const double* __restrict__ rho_n, int64_t nel, double thr,
                                                            GaussTab g3, GaussTab g15, double* __restrict__ vol)
{
    const int64_t e = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (e >= nel) return;
    const int lane = threadIdx.x & 63;
    double xe[4][3], re[4], J[3][3], mn = INFINITY, mx = -INFINITY;
    for (int a = 0; a < 4; ++a) {
        const int64_t n = IEN[e * 4 + a] - 1;
        for (int i = 0; i < 3; ++i) xe[a][i] = X[3 * n + i];
        re[a] = rho_n[n];
        mn = fmin(mn, re[a]);
        mx = fmax(mx, re[a]);
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int a = 0; a < 4; ++a) {
                const double dn = (a == c) ? 1.0 : ((a == 3) ? -1.0 : 0.0);
                s += xe[a][r] * dn;
            }
            J[r][c] = s;
        }
    const double adet = fabs(det3(J));
    double acc = 0.0;
    if (!(mx < thr)) {
        const bool whole = mn >= thr;
        const int n = whole ? 3 : 15;
        const GaussTab& g = whole ? g3 : g15;
        for (int p = lane; p < n * n * n; p += 64) {
            const int i = p % n, j = (p / n) % n, k = p / (n * n);
            const double xi = (g.gp[i] + 1.0) / 2.0;
            const double eta = (g.gp[j] + 1.0) / 2.0 * (1.0 - xi);
            const double zeta = (g.gp[k] + 1.0) / 2.0 * (1.0 - xi - eta);
            if (xi < 0 || eta < 0 || zeta < 0 || xi + eta + zeta > 1.0) continue;
            if (!whole) {   // N = [xi, eta, zeta, 1 - xi - eta - zeta] (ShapeFunctions.jl:53-72)
                const double v = xi * re[0] + eta * re[1] + zeta * re[2] + (1.0 - xi - eta - zeta) * re[3];
                if (v < thr) continue;
            }
            const double jt = (1.0 - xi) * (1.0 - xi) * (1.0 - xi - eta) / 8.0;
            acc += g.gw[i] * g.gw[j] * g.gw[k] * adet * jt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) vol[e] = acc;
}

// sums v[i] and v[i]*s[i] (s may be null) in a fixed order: grid-stride partials, block tree
__global__ void __launch_bounds__(256) sum2_kernel(const double* __restrict__ v, const double* __restrict__ s, int64_t n,
                                                  double* __restrict__ partial)
{
    __shared__ double ra[256], rb[256];
    double a = 0.0, b = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        a += v[i];
        if (s) b += v[i] * s[i];
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) { ra[threadIdx.x] += ra[threadIdx.x + st]; rb[threadIdx.x] += rb[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = ra[0]; partial[2 * blockIdx.x + 1] = rb[0]; }
}

static int sum2(const double* v, const double* s, int64_t n, DevBuf& part, double out[2])
{
    const int nb = 256;
    ENSURE(part, sizeof(double) * 2 * nb);
    sum2_kernel<<<nb, 256>>>(v, s, n, part.as<double>());
    std::vector<double> h(2 * nb);
    HIP_TRY(hipMemcpy(h.data(), part.p, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost));
    out[0] = out[1] = 0.0;
    for (int i = 0; i < nb; ++i) { out[0] += h[2 * i]; out[1] += h[2 * i + 1]; }
    return 0;
}

// ------------------------------------------------------------------------------------
// DenseInNodes: 1 thread / node over its (ascending) element list
// ------------------------------------------------------------------------------------
__global__ void centroid_kernel(const double* __restrict__ X, const int64_t* __restrict__ IEN, int64_t nel, int nen,
                                double* __restrict__ C)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    for (int i = 0; i < 3; ++i) {   // GeometricCentre (NodalDensities.jl:71-80): mean over the element nodes
        double s = 0.0;
        for (int a = 0; a < nen; ++a) s += X[3 * (IEN[e * nen + a] - 1) + i];
        C[3 * e + i] = s / (double)nen;
    }
}

// cyclic Jacobi on a symmetric 4x4 (stands in for LAPACK `eigen`, NodalDensities.jl:159)
__device__ void jacobi4(double A[4][4], double w[4], double V[4][4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        if (off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double apq = A[p][q];
                if (apq != 0.0) {
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                    const double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double akp = A[k][p], akq = A[k][q];
                        A[k][p] = cs * akp - sn * akq;
                        A[k][q] = sn * akp + cs * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double apk = A[p][k], aqk = A[q][k];
                        A[p][k] = cs * apk - sn * aqk;
                        A[q][k] = sn * apk + cs * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double vkp = V[k][p], vkq = V[k][q];
                        V[k][p] = cs * vkp - sn * vkq;
                        V[k][q] = sn * vkp + cs * vkq;
                    }
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = A[i][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // selection sort, ascending
        int m = i;
#pragma unroll
        for (int j = i + 1; j < 4; ++j)
            if (w[j] < w[m]) m = j;
#pragma unroll
        for (int c = i + 1; c < 4; ++c)
            if (c == m) {
                double t = w[i]; w[i] = w[c]; w[c] = t;
#pragma unroll
                for (int k = 0; k < 4; ++k) { t = V[k][i]; V[k][i] = V[k][c]; V[k][c] = t; }
            }
    }
}

__global__ void __launch_bounds__(128) dense_in_nodes_kernel(const double* __restrict__ X, int64_t nnp,
                                                            const uint32_t* __restrict__ ptr,
                                                            const uint32_t* __restrict__ ine,
                                                            const double* __restrict__ C,
                                                            const double* __restrict__ rho_e,
                                                            double* __restrict__ rho_n)
{
    const int64_t nd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nd >= nnp) return;
    const uint32_t p0 = ptr[nd], cnt = ptr[nd + 1] - p0;
    const double x0 = X[3 * nd], x1 = X[3 * nd + 1], x2 = X[3 * nd + 2];
    double out = 0.0;
    if (cnt == 1) {
        out = rho_e[ine[p0]];   // NodalDensities.jl:99-100
    } else if (cnt > 1 && cnt < 4) {   // FilterForNodalDensity (:117-136)
        double L[3] = {0, 0, 0}, Lmax = 0.0;
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t e = ine[p0 + j];
            const double d = norm3(x0 - C[3 * e], x1 - C[3 * e + 1], x2 - C[3 * e + 2]);
            if (j == 0) L[0] = d; else if (j == 1) L[1] = d; else L[2] = d;
            if (d > Lmax) Lmax = d;
        }
        Lmax = Lmax * 1.2;
        double dm = 0.0, den = 0.0;
        for (uint32_t j = 0; j < cnt; ++j) {
            const double Lj = (j == 0) ? L[0] : ((j == 1) ? L[1] : L[2]);
            dm += rho_e[ine[p0 + j]] * (1 - Lj / Lmax);
            den += (1 - Lj / Lmax);
        }
        out = dm / den;
    } else if (cnt > 3) {   // NodalDensityLeastSquares (:145-181)
        double A[4][4], Atb[4] = {0, 0, 0, 0}, bsum = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) A[r][c] = 0.0;
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t e = ine[p0 + j];
            const double row[4] = {1.0, C[3 * e], C[3 * e + 1], C[3 * e + 2]};
            const double b = rho_e[e];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int c = 0; c < 4; ++c) A[r][c] += row[r] * row[c];
                Atb[r] += row[r] * b;
            }
            bsum += b;
        }
        double w[4], V[4][4];
        jacobi4(A, w, V);
        // LamReduction (:190-218)
        const double e1 = fabs(w[3] / w[0]), e2 = fabs(w[3] / w[1]), e3 = fabs(w[3] / w[2]);
        int first = -1;
        if (1e7 > e1 && 3e3 > e2) first = 0;
        else if (1e7 < e1 && 3e3 > e2) first = 1;
        else if (1e7 < e1 && 3e3 < e2) first = (3e3 > e3) ? 2 : 3;
        if (first < 0) {
            out = bsum / (double)cnt;
        } else {
            double b1[4], x2v[4], xs[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) s += V[r][c] * Atb[r];
                b1[c] = s;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) x2v[c] = (c >= first) ? b1[c] / w[c] : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < 4; ++c) s += V[r][c] * x2v[c];
                xs[r] = s;
            }
            out = xs[0] + x0 * xs[1] + x1 * xs[2] + x2 * xs[3];
        }
    }
    rho_n[nd] = out;
}

// node -> element CSR in ascending element order (nodeToElementConnectivity, MeshInformations.jl:69-77)
// ---- node -> elements CSR on the device (each node's elements in ascending order, as a sequential fill leaves them:
// the sums of dense_in_nodes_kernel run in that order) ----
__global__ void __launch_bounds__(256) ine_count_kernel(const int64_t* __restrict__ IEN, int64_t total, int64_t nnp,
                                                       uint32_t* __restrict__ cnt, uint32_t* __restrict__ bad)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t n = IEN[t] - 1;
    if (n < 0 || n >= nnp) { *bad = 1u; return; }
    atomicAdd(&cnt[n], 1u);
}

#define R2S_SCAN_TILE 1024
// inclusive scan of one value per thread over a 1024-thread workgroup; `total` = the sum of the workgroup
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* s_wave /* 16 */, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t x = s_wave[w];
        if (w < wave) before += x;
        all += x;
    }
    __syncthreads();
    total = all;
    return v + before;
}

__global__ void __launch_bounds__(R2S_SCAN_TILE) scan_tile_sums_kernel(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t s_wave[16];
    const int64_t i = (int64_t)blockIdx.x * R2S_SCAN_TILE + threadIdx.x;
    uint32_t total;
    (void)block_scan_1024(i < n ? in[i] : 0u, s_wave, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(R2S_SCAN_TILE) scan_sums_kernel(uint32_t* __restrict__ sums, int64_t nb)   // one workgroup, in place, exclusive
{
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (int64_t base = 0; base < nb; base += R2S_SCAN_TILE) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? sums[i] : 0u;
        uint32_t total;
        const uint32_t inc = block_scan_1024(v, s_wave, total);
        if (i < nb) sums[i] = carry + inc - v;
        carry += total;
    }
}

__global__ void __launch_bounds__(R2S_SCAN_TILE) scan_tiles_kernel(const uint32_t* __restrict__ in, int64_t n, const uint32_t* __restrict__ sums,
                                                                 uint32_t* __restrict__ out)
{
    __shared__ uint32_t s_wave[16];
    const int64_t i = (int64_t)blockIdx.x * R2S_SCAN_TILE + threadIdx.x;
    const uint32_t v = i < n ? in[i] : 0u;
    uint32_t total;
    const uint32_t inc = block_scan_1024(v, s_wave, total);
    if (i < n) out[i] = sums[blockIdx.x] + inc - v;
}

__global__ void __launch_bounds__(256) ine_scatter_kernel(const int64_t* __restrict__ IEN, int64_t total, int nen, int64_t nnp,
                                                         const uint32_t* __restrict__ ptr, uint32_t* __restrict__ cursor,
                                                         uint32_t* __restrict__ ine)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t n = IEN[t] - 1;
    if (n < 0 || n >= nnp) return;
    ine[ptr[n] + atomicAdd(&cursor[n], 1u)] = (uint32_t)(t / nen);
}

__global__ void __launch_bounds__(256) ine_sort_kernel(const uint32_t* __restrict__ ptr, uint32_t* __restrict__ ine, int64_t nnp)
{
    const int64_t nd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nd >= nnp) return;
    const uint32_t p0 = ptr[nd], p1 = ptr[nd + 1];
    for (uint32_t i = p0 + 1; i < p1; ++i) {   // (a handful of entries per node)
        const uint32_t v = ine[i];
        uint32_t j = i;
        for (; j > p0 && ine[j - 1] > v; --j) ine[j] = ine[j - 1];
        ine[j] = v;
    }
}

// ws: [0] ptr (nnp + 1), [1] counts / cursors (nnp + 1), [2] ine, [3] tile sums + the flag
static int build_ine_dev(const int64_t* dIEN, int64_t nel, int nen, int64_t nnp, DevBuf* ws)
{
    const int64_t total = nel * nen, np1 = nnp + 1, nb = (np1 + R2S_SCAN_TILE - 1) / R2S_SCAN_TILE;
    if (ws[0].ensure(4 * (size_t)np1) || ws[1].ensure(4 * (size_t)np1) || ws[2].ensure(4 * (size_t)total) || ws[3].ensure(4 * (size_t)(nb + 1)))
        return fail(R2S_ERR_NOMEM, "hipMalloc failed");
    uint32_t* ptr = ws[0].as<uint32_t>();
    uint32_t* cnt = ws[1].as<uint32_t>();
    uint32_t* sums = ws[3].as<uint32_t>();
    uint32_t* bad = sums + nb;
    HIP_TRY(hipMemsetAsync(cnt, 0, 4 * (size_t)np1, nullptr));
    HIP_TRY(hipMemsetAsync(bad, 0, 4, nullptr));
    ine_count_kernel<<<(unsigned)((total + 255) / 256), 256>>>(dIEN, total, nnp, cnt, bad);
    scan_tile_sums_kernel<<<(unsigned)nb, R2S_SCAN_TILE>>>(cnt, np1, sums);
    scan_sums_kernel<<<1, R2S_SCAN_TILE>>>(sums, nb);
    scan_tiles_kernel<<<(unsigned)nb, R2S_SCAN_TILE>>>(cnt, np1, sums, ptr);
    HIP_TRY(hipMemsetAsync(cnt, 0, 4 * (size_t)np1, nullptr));
    ine_scatter_kernel<<<(unsigned)((total + 255) / 256), 256>>>(dIEN, total, nen, nnp, ptr, cnt, ws[2].as<uint32_t>());
    ine_sort_kernel<<<(unsigned)((nnp + 255) / 256), 256>>>(ptr, ws[2].as<uint32_t>(), nnp);
    uint32_t h_bad = 0;
    HIP_TRY(hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost));
    if (h_bad) return fail(R2S_ERR_ARG, "IEN contains node ids outside 1..nnp");
    return 0;
}

struct MeshDev {
    DevBuf X, IEN, a, b;
    void release() { X.release(); IEN.release(); a.release(); b.release(); }
};

static int upload_mesh(MeshDev& m, const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, int nen)
{
    ENSURE(m.X, sizeof(double) * 3 * (size_t)nnp);
    ENSURE(m.IEN, sizeof(int64_t) * (size_t)(nel * nen));
    HIP_TRY(hipMemcpy(m.X.p, X, sizeof(double) * 3 * (size_t)nnp, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m.IEN.p, IEN, sizeof(int64_t) * (size_t)(nel * nen), hipMemcpyHostToDevice));
    return 0;
}

static void tables(GaussTab& g3, GaussTab& g15)
{
    memset(&g3, 0, sizeof g3);
    memset(&g15, 0, sizeof g15);
    r2s_internal_gauss_legendre(3, g3.gp, g3.gw);
    r2s_internal_gauss_legendre(15, g15.gp, g15.gw);
}

// ---- device-pointer forms (current device; synchronous) -------------------------------------------------
namespace r2s_int {

int mesh_volume_dev(const double* dX, const int64_t* dIEN, int64_t nel, int elem_type, const double* d_rho_e,
                    double* V_domain, double* V_frac, DevBuf* ws)
{
    DevBuf own[2];   // (ws: two buffers the caller keeps between calls)
    DevBuf& vol = ws ? ws[0] : own[0];
    DevBuf& part = ws ? ws[1] : own[1];
    GaussTab g3, g15;
    tables(g3, g15);
    auto done = [&](int r) { own[0].release(); own[1].release(); return r; };
    if (vol.ensure(sizeof(double) * (size_t)nel)) return done(fail(R2S_ERR_NOMEM, "hipMalloc failed"));
    if (elem_type == R2S_HEX8)
        elem_volume_kernel<<<(unsigned)((nel + 3) / 4), 256>>>(dX, dIEN, nullptr, nel, 0, 0.0, g3, g15, vol.as<double>());
    else
        tet_volume_kernel<<<(unsigned)((nel + 255) / 256), 256>>>(dX, dIEN, nel, g3, vol.as<double>());
    double s[2];
    int rc = sum2(vol.as<double>(), d_rho_e, nel, part, s);
    if (rc) return done(rc);
    *V_domain = s[0];
    *V_frac = s[1] / s[0];   // MeshVolume.jl:41
    return done(0);
}

int dense_in_nodes_dev(const double* dX, int64_t nnp, const int64_t* dIEN, int64_t nel, int elem_type, const double* d_rho_e,
                       double* d_rho_n_out, DevBuf* ws)
{
    const int nen = elem_type == R2S_HEX8 ? 8 : 4;
    DevBuf own[5];   // (ws: five buffers the caller keeps between calls)
    DevBuf* b = ws ? ws : own;
    auto done = [&](int r) { for (DevBuf& x : own) x.release(); return r; };
    int rc = build_ine_dev(dIEN, nel, nen, nnp, b);
    if (rc) return done(rc);
    if (b[4].ensure(sizeof(double) * 3 * (size_t)nel)) return done(fail(R2S_ERR_NOMEM, "hipMalloc failed"));
    centroid_kernel<<<(unsigned)((nel + 255) / 256), 256>>>(dX, dIEN, nel, nen, b[4].as<double>());
    dense_in_nodes_kernel<<<(unsigned)((nnp + 127) / 128), 128>>>(dX, nnp, b[0].as<uint32_t>(), b[2].as<uint32_t>(), b[4].as<double>(), d_rho_e,
                                                                 d_rho_n_out);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return done(fail(R2S_ERR_HIP, "dense_in_nodes kernels failed: %s", hipGetErrorString(e)));
    return done(0);
}

struct IsoVolume {   // calculate_isocontour_volume (:1-75) for a sequence of thresholds
    const double* dX;
    const int64_t* dIEN;
    int64_t nel;
    int elem_type;
    const double* d_rho_n;
    DevBuf vol, part;
    GaussTab g3, g15;
    int init()
    {
        tables(g3, g15);
        if (vol.ensure(sizeof(double) * (size_t)nel)) return fail(R2S_ERR_NOMEM, "hipMalloc failed");
        return 0;
    }
    int run(double thr, double* out)
    {
        if (elem_type == R2S_HEX8)
            elem_volume_kernel<<<(unsigned)((nel + 3) / 4), 256>>>(dX, dIEN, d_rho_n, nel, 1, thr, g3, g15, vol.as<double>());
        else
            tet_iso_volume_kernel<<<(unsigned)((nel + 3) / 4), 256>>>(dX, dIEN, d_rho_n, nel, thr, g3, g15, vol.as<double>());
        double s[2] = {0, 0};
        int rc = sum2(vol.as<double>(), nullptr, nel, part, s);
        *out = s[0];
        return rc;
    }
    void release() { vol.release(); part.release(); }
};

int isocontour_volume_dev(const double* dX, const int64_t* dIEN, int64_t nel, int elem_type, const double* d_rho_n,
                          double thr, double* volume_out)
{
    IsoVolume iv{dX, dIEN, nel, elem_type, d_rho_n};
    int rc = iv.init();
    if (!rc) rc = iv.run(thr, volume_out);
    iv.release();
    return rc;
}

int find_threshold_dev(const double* dX, const int64_t* dIEN, int64_t nel, int elem_type, const double* d_rho_n,
                       double target_volume, double tol, int maxit, double* rho_t_out, int* iters_out)
{
    IsoVolume iv{dX, dIEN, nel, elem_type, d_rho_n};
    int rc = iv.init();
    auto done = [&](int r) { iv.release(); return r; };
    if (rc) return done(rc);
    double lo = 0.0, hi = 1.0, vmin = 0.0, vmax = 0.0;
    if ((rc = iv.run(hi, &vmin)) || (rc = iv.run(lo, &vmax))) return done(rc);
    if (target_volume > vmax || target_volume < vmin)   // Isocontour_volume.jl:93-95
        return done(fail(R2S_ERR_ARG, "Requested volume %.17g is outside the possible range [%.17g, %.17g]",
                         target_volume, vmin, vmax));
    int it = 0;
    double best = 0.0, best_err = INFINITY;
    while (it < maxit) {
        const double thr = (lo + hi) / 2;
        double v = 0.0;
        if ((rc = iv.run(thr, &v))) return done(rc);
        const double e = std::fabs(v - target_volume) / target_volume;
        if (e < best_err) { best = thr; best_err = e; }
        if (e < tol) break;
        if (v > target_volume) lo = thr; else hi = thr;
        it++;
    }
    *rho_t_out = best;
    if (iters_out) *iters_out = it;
    return done(0);
}

}  // namespace r2s_int

extern "C" {

int r2s_mesh_volume(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, int32_t elem_type,
                    const double* rho_e, int32_t device, double* V_domain, double* V_frac)
{
    if (!X || !IEN || !rho_e || !V_domain || !V_frac || nnp <= 0 || nel <= 0) return fail(R2S_ERR_ARG, "bad argument");
    if (elem_type != R2S_HEX8 && elem_type != R2S_TET4) return fail(R2S_ERR_UNSUPPORTED, "unknown element type %d", elem_type);
    const int nen = elem_type == R2S_HEX8 ? 8 : 4;
    int rc = use_device(device);
    if (rc) return rc;
    MeshDev m;
    DevBuf rho;
    auto done = [&](int r) { m.release(); rho.release(); return r; };
    if ((rc = upload_mesh(m, X, nnp, IEN, nel, nen))) return done(rc);
    if (rho.ensure(sizeof(double) * (size_t)nel)) return done(fail(R2S_ERR_NOMEM, "hipMalloc failed"));
    if (hipMemcpy(rho.p, rho_e, sizeof(double) * (size_t)nel, hipMemcpyHostToDevice) != hipSuccess)
        return done(fail(R2S_ERR_HIP, "hipMemcpy failed"));
    return done(r2s_int::mesh_volume_dev(m.X.as<double>(), m.IEN.as<int64_t>(), nel, elem_type, rho.as<double>(), V_domain, V_frac));
}

int r2s_dense_in_nodes(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, int32_t elem_type,
                       const double* rho_e, int32_t device, double* rho_n_out)
{
    if (!X || !IEN || !rho_e || !rho_n_out || nnp <= 0 || nel <= 0) return fail(R2S_ERR_ARG, "bad argument");
    if (elem_type != R2S_HEX8 && elem_type != R2S_TET4) return fail(R2S_ERR_UNSUPPORTED, "unknown element type %d", elem_type);
    const int nen = elem_type == R2S_HEX8 ? 8 : 4;
    int rc = use_device(device);
    if (rc) return rc;
    MeshDev m;
    DevBuf rho, out;
    auto done = [&](int r) { m.release(); rho.release(); out.release(); return r; };
    if ((rc = upload_mesh(m, X, nnp, IEN, nel, nen))) return done(rc);
    if (rho.ensure(sizeof(double) * (size_t)nel) || out.ensure(sizeof(double) * (size_t)nnp))
        return done(fail(R2S_ERR_NOMEM, "hipMalloc failed"));
    if (hipMemcpy(rho.p, rho_e, sizeof(double) * (size_t)nel, hipMemcpyHostToDevice) != hipSuccess)
        return done(fail(R2S_ERR_HIP, "hipMemcpy failed"));
    if ((rc = r2s_int::dense_in_nodes_dev(m.X.as<double>(), nnp, m.IEN.as<int64_t>(), nel, elem_type, rho.as<double>(),
                                          out.as<double>())))
        return done(rc);
    if (hipMemcpy(rho_n_out, out.p, sizeof(double) * (size_t)nnp, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(R2S_ERR_HIP, "copy failed: %s", hipGetErrorString(hipGetLastError())));
    return done(0);
}

static int upload_nodal(MeshDev& m, DevBuf& rho, const double* X, int64_t nnp, const int64_t* IEN, int64_t nel,
                        int32_t elem_type, const double* rho_n)
{
    const int nen = elem_type == R2S_HEX8 ? 8 : 4;
    int rc = upload_mesh(m, X, nnp, IEN, nel, nen);
    if (rc) return rc;
    ENSURE(rho, sizeof(double) * (size_t)nnp);
    HIP_TRY(hipMemcpy(rho.p, rho_n, sizeof(double) * (size_t)nnp, hipMemcpyHostToDevice));
    return 0;
}

int r2s_find_threshold(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, const double* rho_n,
                       double target_volume, double tol, int32_t maxit, int32_t device, double* rho_t_out,
                       int32_t* iters_out)
{
    return r2s_find_threshold_et(X, nnp, IEN, nel, R2S_HEX8, rho_n, target_volume, tol, maxit, device, rho_t_out, iters_out);
}

int r2s_find_threshold_et(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, int32_t elem_type,
                          const double* rho_n, double target_volume, double tol, int32_t maxit, int32_t device,
                          double* rho_t_out, int32_t* iters_out)
{
    if (!X || !IEN || !rho_n || !rho_t_out || nnp <= 0 || nel <= 0) return fail(R2S_ERR_ARG, "bad argument");
    if (elem_type != R2S_HEX8 && elem_type != R2S_TET4) return fail(R2S_ERR_UNSUPPORTED, "unknown element type %d", elem_type);
    int rc = use_device(device);
    if (rc) return rc;
    MeshDev m;
    DevBuf rho;
    auto done = [&](int r) { m.release(); rho.release(); return r; };
    if ((rc = upload_nodal(m, rho, X, nnp, IEN, nel, elem_type, rho_n))) return done(rc);
    int it = 0;
    rc = r2s_int::find_threshold_dev(m.X.as<double>(), m.IEN.as<int64_t>(), nel, elem_type, rho.as<double>(), target_volume,
                                     tol, maxit, rho_t_out, &it);
    if (iters_out) *iters_out = it;
    return done(rc);
}

int r2s_isocontour_volume(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, int32_t elem_type,
                          const double* rho_n, double threshold, int32_t device, double* volume_out)
{
    if (!X || !IEN || !rho_n || !volume_out || nnp <= 0 || nel <= 0) return fail(R2S_ERR_ARG, "bad argument");
    if (elem_type != R2S_HEX8 && elem_type != R2S_TET4) return fail(R2S_ERR_UNSUPPORTED, "unknown element type %d", elem_type);
    int rc = use_device(device);
    if (rc) return rc;
    MeshDev m;
    DevBuf rho;
    auto done = [&](int r) { m.release(); rho.release(); return r; };
    if ((rc = upload_nodal(m, rho, X, nnp, IEN, nel, elem_type, rho_n))) return done(rc);
    return done(r2s_int::isocontour_volume_dev(m.X.as<double>(), m.IEN.as<int64_t>(), nel, elem_type, rho.as<double>(),
                                               threshold, volume_out));
}

}  // extern "C"
