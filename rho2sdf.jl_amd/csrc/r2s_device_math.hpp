// Device-side element math for the signed-distance hot path (gfx950, FP64 VALU).
//
// Every function states which reference function it replaces.  Expressions that
// feed integer / boolean decisions keep the reference's operation order; the
// translation unit is compiled with -ffp-contract=off so nothing is contracted.
// All small arrays are indexed with compile-time constants after unrolling so
// they live in VGPRs (no scratch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace r2s {

#define R2S_DEV __device__ __forceinline__

// One pre-gathered HEX8 element: X[:, IEN[:, el]] and rho_n[IEN[:, el]] plus the
// element AABB (SignDetection.jl:16-21) and max nodal density (SignDetection.jl:33-36).
struct alignas(16) ElemRec {
    double X[8][3];
    double r[8];
    double mn[3];
    double mx[3];
    double rmax;
    double rmin;
    // monomial coefficients of the trilinear maps, used by the restated solvers only:
    // X(xi) = sum_m C[m] mono_m(xi), mono = [1, x1, x2, x3, x1x2, x1x3, x2x3, x1x2x3]
    double C[8][3];
    double Cr[8];
    // six half-spaces pn[f].x <= po[f] that contain the image of the cube |xi| <= 1.011 (convex hull of
    // the inflated corners): a point outside one of them has no local coordinates with max|xi| < 1.01, so
    // Sign_Detection's candidate test cannot accept it and the Newton solve is skipped (exact pruning)
    double pn[6][3];
    double po[6];
    // the same six normals with INNER offsets: pn[f].x <= pi[f] for all f describes a convex region that lies inside the
    // element (every bilinear face patch lies in the hull of its four corners, i.e. beyond its inner plane; the region
    // holds the image of xi = 0 and cannot cross a face).  A lattice point in it has local coordinates with
    // max|xi| <= 1 without any Newton solve (sign_project_kernel uses it for elements whose nodal densities all exceed
    // rho_t).  -inf: no inner region (degenerate / inverted element).
    double pi[6];
    // first Newton step of the inverse map (xi = 0): rows of the adjugate of J(0) = [C1 C2 C3] and 1/det
    double nw[10];
};

// band work item: one boundary-face triangle (process_triangle_projection!,
// sdfOnDensityField.jl:628-815) or one iso-surface projection
// (process_isocontour_element!, :606-624), with its mini-AABB cell box
// (calculateMiniAABB_grid, Grid.jl:122-154).
struct alignas(16) BandItem {
    int32_t imin[3];
    int32_t imax[3];
    int32_t el;    // 0-based element
    int32_t kind;  // 0 iso projection, 1 triangle of a solid element, 2 triangle of an iso element
    int32_t im;    // row of the barycentric system replaced by ones (TriangularMeshUtils.jl:19-21)
    int32_t p0, p1;  // partial-pivot row swaps of the 3x3 LU
    int32_t sing;  // LU hit a zero pivot
    // iso items only: lattice box (voxels whose cell can lie in the cell box, clipped to the
    // Z-slab) that the item-major projection kernel sweeps, and its result slots
    int32_t lo[3];
    int32_t dim[3];      // 0 in any axis = empty
    uint32_t chunk_off;  // first 64-voxel WORK chunk of this item (dense enumeration of the box)
    uint32_t store_off;  // first 64-slot STORAGE chunk: results are stored per 4x4x4 voxel tile of the box
    double tri[3][3];  // vertices x1,x2,x3
    double n[3];       // unit normal
    double L[3];       // edge lengths
    double eh[3][3];   // unit edge directions Et[j]/L
    double lu[3][3];   // LU factors of the barycentric matrix (L unit-lower, U upper)
};

struct GridDev {
    double amin[3];
    double amax[3];
    double cell;
    int32_t N[3];
    int32_t nx, ny, nz;  // N+1
};

// ---- shape functions (src/ShapeFunctions/hex8_shape.jl:2-108) -------------------
R2S_DEV void hex8_shape(const double xi[3], double N[8])
{
    double x1m = xi[0] - 1, x1p = xi[0] + 1, x2m = xi[1] - 1, x2p = xi[1] + 1;
    double x3m = xi[2] - 1, x3p = xi[2] + 1;
    double t1 = x1m * x2m, t2 = x1p * x2m, t3 = x1p * x2p, t4 = x1m * x2p;
    const double c = 0.125;
    N[0] = -c * t1 * x3m;
    N[1] = c * t2 * x3m;
    N[2] = -c * t3 * x3m;
    N[3] = c * t4 * x3m;
    N[4] = c * t1 * x3p;
    N[5] = -c * t2 * x3p;
    N[6] = c * t3 * x3p;
    N[7] = -c * t4 * x3p;
}

R2S_DEV void hex8_shape_d(const double xi[3], double N[8], double dN[8][3])
{
    double x1m = xi[0] - 1, x1p = xi[0] + 1, x2m = xi[1] - 1, x2p = xi[1] + 1;
    double x3m = xi[2] - 1, x3p = xi[2] + 1;
    double t1 = x1m * x2m, t2 = x1p * x2m, t3 = x1p * x2p, t4 = x1m * x2p;
    const double c = 0.125;
    N[0] = -c * t1 * x3m;
    N[1] = c * t2 * x3m;
    N[2] = -c * t3 * x3m;
    N[3] = c * t4 * x3m;
    N[4] = c * t1 * x3p;
    N[5] = -c * t2 * x3p;
    N[6] = c * t3 * x3p;
    N[7] = -c * t4 * x3p;
    double d1 = c * x3m, d1p = c * x3p;
    dN[0][0] = -d1 * x2m;  dN[1][0] = d1 * x2m;   dN[2][0] = -d1 * x2p;  dN[3][0] = d1 * x2p;
    dN[4][0] = d1p * x2m;  dN[5][0] = -d1p * x2m; dN[6][0] = d1p * x2p;  dN[7][0] = -d1p * x2p;
    dN[0][1] = -d1 * x1m;  dN[1][1] = d1 * x1p;   dN[2][1] = -d1 * x1p;  dN[3][1] = d1 * x1m;
    dN[4][1] = d1p * x1m;  dN[5][1] = -d1p * x1p; dN[6][1] = d1p * x1p;  dN[7][1] = -d1p * x1m;
    dN[0][2] = -c * t1;    dN[1][2] = c * t2;     dN[2][2] = -c * t3;    dN[3][2] = c * t4;
    dN[4][2] = c * t1;     dN[5][2] = -c * t2;    dN[6][2] = c * t3;     dN[7][2] = -c * t4;
}

R2S_DEV double norm3(double a, double b, double c) { return sqrt(a * a + b * b + c * c); }

// grid point (Grid.jl:87) and its cell index (Grid.jl:58)
R2S_DEV double grid_coord(const GridDev& g, int ax, int i) { return g.amin[ax] + g.cell * (double)i; }
R2S_DEV double cell_of(const GridDev& g, int ax, double x)
{
    return floor((double)g.N[ax] * (x - g.amin[ax]) / (g.amax[ax] - g.amin[ax]));
}

// ---- inverse isoparametric map (FindLocalCoordinates.jl:16-107) ------------------
// Box-clamped Newton from xi = 0 on Xe N(xi) = x; see DESIGN.md "inverse map".
// signs of the monomial expansion: C[m] = 1/8 sum_k MONO_SIGN[m][k] X_k (node order of hex8_shape.jl)
__device__ const double c_mono_sign[8][8] = {
    {1, 1, 1, 1, 1, 1, 1, 1},     {-1, 1, 1, -1, -1, 1, 1, -1}, {-1, -1, 1, 1, -1, -1, 1, 1},
    {-1, -1, -1, -1, 1, 1, 1, 1}, {1, -1, 1, -1, 1, -1, 1, -1}, {1, -1, -1, 1, -1, 1, 1, -1},
    {1, 1, -1, -1, -1, -1, 1, 1}, {-1, 1, -1, 1, 1, -1, 1, -1}};

R2S_DEV void hex8_monomials(ElemRec& R)
{
    for (int m = 0; m < 8; ++m) {
        for (int i = 0; i < 3; ++i) {
            double t = 0.0;
            for (int k = 0; k < 8; ++k) t += c_mono_sign[m][k] * R.X[k][i];
            R.C[m][i] = 0.125 * t;
        }
        double t = 0.0;
        for (int k = 0; k < 8; ++k) t += c_mono_sign[m][k] * R.r[k];
        R.Cr[m] = 0.125 * t;
    }
}

// Bounding half-spaces of the inflated element (see ElemRec::pn).  Face f = 2a + (s > 0): normal = cross
// product of the two in-face tangents at the face centre, oriented along s * dX/dxi_a; offset = max of
// n.X over the 8 corners of the cube |xi| = 1.011 (a trilinear map takes the cube into the convex hull of
// the corner images) plus a rounding margin.  Degenerate faces give n = 0 and never reject.
R2S_DEV void hex8_plane(const ElemRec& R, int a, int sg, double n[3], double& po, double& pin)
{
    const double lamb = 1.011;
    double ext = 0.0;
    for (int i = 0; i < 3; ++i) ext += R.mx[i] - R.mn[i];
    const double s = sg ? 1.0 : -1.0;
    double xi[3] = {0.0, 0.0, 0.0};
    xi[a] = s;
    double J[3][3];   // J[i][q] = dX_i / dxi_q
    for (int i = 0; i < 3; ++i) {
        J[i][0] = R.C[1][i] + xi[1] * R.C[4][i] + xi[2] * R.C[5][i] + xi[1] * xi[2] * R.C[7][i];
        J[i][1] = R.C[2][i] + xi[0] * R.C[4][i] + xi[2] * R.C[6][i] + xi[0] * xi[2] * R.C[7][i];
        J[i][2] = R.C[3][i] + xi[0] * R.C[5][i] + xi[1] * R.C[6][i] + xi[0] * xi[1] * R.C[7][i];
    }
    const int b = (a + 1) % 3, c = (a + 2) % 3;
    n[0] = J[1][b] * J[2][c] - J[2][b] * J[1][c];
    n[1] = J[2][b] * J[0][c] - J[0][b] * J[2][c];
    n[2] = J[0][b] * J[1][c] - J[1][b] * J[0][c];
    const double along = s * (n[0] * J[0][a] + n[1] * J[1][a] + n[2] * J[2][a]);
    if (along < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    if (!(along != 0.0)) { n[0] = n[1] = n[2] = 0.0; }   // degenerate or NaN: never rejects
    double off = -INFINITY;
    for (int k = 0; k < 8; ++k) {
        const double s1 = (k & 1) ? lamb : -lamb, s2 = (k & 2) ? lamb : -lamb, s3 = (k & 4) ? lamb : -lamb;
        double v = 0.0;
        for (int i = 0; i < 3; ++i) {
            const double Xi = R.C[0][i] + s1 * R.C[1][i] + s2 * R.C[2][i] + s3 * R.C[3][i] +
                              s1 * s2 * R.C[4][i] + s1 * s3 * R.C[5][i] + s2 * s3 * R.C[6][i] +
                              s1 * s2 * s3 * R.C[7][i];
            v += n[i] * Xi;
        }
        off = fmax(off, v);
    }
    const double nn = fabs(n[0]) + fabs(n[1]) + fabs(n[2]);
    po = off + 1e-9 * nn * ext;
    // inner offset: min of n.X over the four corners of the face xi_a = s, minus the rounding margin; valid only when the
    // element is not inverted anywhere (Jacobian determinant of one sign at the centre and the eight corners) and its
    // centre lies strictly inside this half-space
    double inner = INFINITY;
    for (int k = 0; k < 4; ++k) {
        double c[3];
        c[a] = s;
        c[(a + 1) % 3] = (k & 1) ? 1.0 : -1.0;
        c[(a + 2) % 3] = (k & 2) ? 1.0 : -1.0;
        double v = 0.0;
        for (int i = 0; i < 3; ++i) {
            const double Xi = R.C[0][i] + c[0] * R.C[1][i] + c[1] * R.C[2][i] + c[2] * R.C[3][i] + c[0] * c[1] * R.C[4][i] +
                              c[0] * c[2] * R.C[5][i] + c[1] * c[2] * R.C[6][i] + c[0] * c[1] * c[2] * R.C[7][i];
            v += n[i] * Xi;
        }
        inner = fmin(inner, v);
    }
    inner -= 1e-9 * nn * ext;
    bool ok = along != 0.0 && nn > 0.0;
    double sgn = 0.0;
    for (int k = 0; k < 9 && ok; ++k) {   // k = 8: the centre
        const double c0 = k == 8 ? 0.0 : ((k & 1) ? 1.0 : -1.0), c1 = k == 8 ? 0.0 : ((k & 2) ? 1.0 : -1.0),
                     c2 = k == 8 ? 0.0 : ((k & 4) ? 1.0 : -1.0);
        double Jc[3][3];
        for (int i = 0; i < 3; ++i) {
            Jc[i][0] = R.C[1][i] + c1 * R.C[4][i] + c2 * R.C[5][i] + c1 * c2 * R.C[7][i];
            Jc[i][1] = R.C[2][i] + c0 * R.C[4][i] + c2 * R.C[6][i] + c0 * c2 * R.C[7][i];
            Jc[i][2] = R.C[3][i] + c0 * R.C[5][i] + c1 * R.C[6][i] + c0 * c1 * R.C[7][i];
        }
        const double det = Jc[0][0] * (Jc[1][1] * Jc[2][2] - Jc[1][2] * Jc[2][1]) - Jc[0][1] * (Jc[1][0] * Jc[2][2] - Jc[1][2] * Jc[2][0]) +
                           Jc[0][2] * (Jc[1][0] * Jc[2][1] - Jc[1][1] * Jc[2][0]);
        if (!(det != 0.0)) ok = false;
        else if (sgn == 0.0) sgn = det;
        else if ((det > 0.0) != (sgn > 0.0)) ok = false;
    }
    const double centre = n[0] * R.C[0][0] + n[1] * R.C[0][1] + n[2] * R.C[0][2];
    if (!(centre < inner)) ok = false;
    pin = ok ? inner : -INFINITY;
}

// constants of the inverse map's first Newton step (see inv_map_hex8): the cofactor / det / reciprocal
// expressions of its general iteration, evaluated at xi = 0 where J[i][q] = C[q+1][i]
R2S_DEV void hex8_newton0(ElemRec& R)
{
    double J[3][3];
    for (int i = 0; i < 3; ++i)
        for (int q = 0; q < 3; ++q) J[i][q] = R.C[q + 1][i];
    const double c00 = fma(J[1][1], J[2][2], -(J[1][2] * J[2][1]));
    const double c01 = fma(J[1][2], J[2][0], -(J[1][0] * J[2][2]));
    const double c02 = fma(J[1][0], J[2][1], -(J[1][1] * J[2][0]));
    const double det = fma(J[0][2], c02, fma(J[0][1], c01, J[0][0] * c00));   // = dot3(J[0][:], c0:)
    const double c10 = fma(J[0][2], J[2][1], -(J[0][1] * J[2][2]));
    const double c11 = fma(J[0][0], J[2][2], -(J[0][2] * J[2][0]));
    const double c12 = fma(J[0][1], J[2][0], -(J[0][0] * J[2][1]));
    const double c20 = fma(J[0][1], J[1][2], -(J[0][2] * J[1][1]));
    const double c21 = fma(J[0][2], J[1][0], -(J[0][0] * J[1][2]));
    const double c22 = fma(J[0][0], J[1][1], -(J[0][1] * J[1][0]));
    R.nw[0] = c00; R.nw[1] = c10; R.nw[2] = c20;
    R.nw[3] = c01; R.nw[4] = c11; R.nw[5] = c21;
    R.nw[6] = c02; R.nw[7] = c12; R.nw[8] = c22;
    R.nw[9] = 1.0 / det;
}

// true when x lies outside one of the bounding half-spaces (no local coordinates with max|xi| < 1.01)
template <class ER>
R2S_DEV bool hex8_outside(const ER& E, const double x[3])
{
    bool out = false;
#pragma unroll
    for (int f = 0; f < 6; ++f)
        out = out || (E.pn[f][0] * x[0] + E.pn[f][1] * x[1] + E.pn[f][2] * x[2] > E.po[f]);
    return out;
}

// true when x lies in the inner region (ElemRec::pi): inside the element, max|xi| <= 1, no Newton solve needed
template <class ER>
R2S_DEV bool hex8_inner(const ER& E, const double x[3])
{
    bool in = true;
#pragma unroll
    for (int f = 0; f < 6; ++f)
        in = in && (E.pn[f][0] * x[0] + E.pn[f][1] * x[1] + E.pn[f][2] * x[2] <= E.pi[f]);
    return in;
}

// value, gradient and mixed second derivatives of one scalar trilinear field (12 FMAs)
struct TriEval {
    double v, d1, d2, d3, m12, m13, m23;
};
R2S_DEV TriEval tri_eval_full(double c0, double c1, double c2, double c3, double c12, double c13, double c23,
                              double c123, const double xi[3])
{
    TriEval o;
    const double q0 = fma(xi[2], c3, c0);
    const double q1 = fma(xi[2], c13, c1);
    const double q2 = fma(xi[2], c23, c2);
    const double q3 = fma(xi[2], c123, c12);
    const double r0 = fma(xi[1], q2, q0);
    const double r1 = fma(xi[1], q3, q1);
    o.v = fma(xi[0], r1, r0);
    o.d1 = r1;
    o.d2 = fma(xi[0], q3, q2);
    o.m13 = fma(xi[1], c123, c13);
    o.m23 = fma(xi[0], c123, c23);
    o.d3 = fma(xi[0], o.m13, fma(xi[1], c23, c3));
    o.m12 = q3;
    return o;
}
R2S_DEV double tri_eval_value(double c0, double c1, double c2, double c3, double c12, double c13, double c23,
                              double c123, const double xi[3])
{
    const double q0 = fma(xi[2], c3, c0);
    const double q1 = fma(xi[2], c13, c1);
    const double q2 = fma(xi[2], c23, c2);
    const double q3 = fma(xi[2], c123, c12);
    return fma(xi[0], fma(xi[1], q3, q1), fma(xi[1], q2, q0));
}
#define R2S_CX(E, i) (E).C[0][i], (E).C[1][i], (E).C[2][i], (E).C[3][i], (E).C[4][i], (E).C[5][i], (E).C[6][i], (E).C[7][i]
#define R2S_CR(E) (E).Cr[0], (E).Cr[1], (E).Cr[2], (E).Cr[3], (E).Cr[4], (E).Cr[5], (E).Cr[6], (E).Cr[7]

// a0 b0 + a1 b1 + a2 b2 in a fixed fused order (one multiply, two FMAs) - the oracle evaluates the same
// IEEE operations, so the solvers below stay bit-identical to it
R2S_DEV double dot3(double a0, double a1, double a2, double b0, double b1, double b2)
{
    return fma(a2, b2, fma(a1, b1, a0 * b0));
}

#define R2S_INV_TOL 1e-7   // step tolerance of the inverse map = the oracle's INV_TOL (see the note there)
template <class ER>
R2S_DEV bool inv_map_hex8(const ER& E, const double x[3], double xi[3])
{
    xi[0] = xi[1] = xi[2] = 0.0;
    {
        // iteration 0: at xi = 0 the map gives X = C[0] and J = [C1 C2 C3] exactly, so the adjugate and
        // 1/det of that first Newton step are per-element constants (ElemRec::nw, hex8_newton0) - the same
        // IEEE operations as the general iteration below, one sixth of its cost
        const double R0 = E.C[0][0] - x[0], R1 = E.C[0][1] - x[1], R2 = E.C[0][2] - x[2];
        const double d0 = -dot3(E.nw[0], E.nw[1], E.nw[2], R0, R1, R2) * E.nw[9];
        const double d1 = -dot3(E.nw[3], E.nw[4], E.nw[5], R0, R1, R2) * E.nw[9];
        const double d2 = -dot3(E.nw[6], E.nw[7], E.nw[8], R0, R1, R2) * E.nw[9];
        const double n0 = fmin(fmax(xi[0] + d0, -1.1), 1.1);
        const double n1 = fmin(fmax(xi[1] + d1, -1.1), 1.1);
        const double n2 = fmin(fmax(xi[2] + d2, -1.1), 1.1);
        const double step = fmax(fabs(n0 - xi[0]), fmax(fabs(n1 - xi[1]), fabs(n2 - xi[2])));
        xi[0] = n0; xi[1] = n1; xi[2] = n2;
        if (!(step > R2S_INV_TOL)) {
            if (step != step) { xi[0] = xi[1] = xi[2] = 10.0; return false; }
            return true;
        }
    }
    for (int it = 1; it < 50; ++it) {
        double R[3], J[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const TriEval t = tri_eval_full(R2S_CX(E, i), xi);
            R[i] = t.v - x[i];
            J[i][0] = t.d1; J[i][1] = t.d2; J[i][2] = t.d3;
        }
        double c00 = fma(J[1][1], J[2][2], -(J[1][2] * J[2][1]));
        double c01 = fma(J[1][2], J[2][0], -(J[1][0] * J[2][2]));
        double c02 = fma(J[1][0], J[2][1], -(J[1][1] * J[2][0]));
        double det = dot3(J[0][0], J[0][1], J[0][2], c00, c01, c02);
        double c10 = fma(J[0][2], J[2][1], -(J[0][1] * J[2][2]));
        double c11 = fma(J[0][0], J[2][2], -(J[0][2] * J[2][0]));
        double c12 = fma(J[0][1], J[2][0], -(J[0][0] * J[2][1]));
        double c20 = fma(J[0][1], J[1][2], -(J[0][2] * J[1][1]));
        double c21 = fma(J[0][2], J[1][0], -(J[0][0] * J[1][2]));
        double c22 = fma(J[0][0], J[1][1], -(J[0][1] * J[1][0]));
        double rdet = 1.0 / det;
        double d0 = -dot3(c00, c10, c20, R[0], R[1], R[2]) * rdet;
        double d1 = -dot3(c01, c11, c21, R[0], R[1], R[2]) * rdet;
        double d2 = -dot3(c02, c12, c22, R[0], R[1], R[2]) * rdet;
        double n0 = fmin(fmax(xi[0] + d0, -1.1), 1.1);
        double n1 = fmin(fmax(xi[1] + d1, -1.1), 1.1);
        double n2 = fmin(fmax(xi[2] + d2, -1.1), 1.1);
        double step = fmax(fabs(n0 - xi[0]), fmax(fabs(n1 - xi[1]), fabs(n2 - xi[2])));
        xi[0] = n0; xi[1] = n1; xi[2] = n2;
        if (!(step > R2S_INV_TOL)) {
            if (step != step) break;
            return true;
        }
    }
    xi[0] = xi[1] = xi[2] = 10.0;
    return false;
}

// ---- projection onto the iso-surface inside one HEX8 ----------------------------
// Replaces compute_coords_on_iso (ComputeCoordsOnIso.jl:16-87, NLopt SLSQP):
// same objective / constraint / bounds / start; SQP described in DESIGN.md.
struct QpOut {
    double d[3];
    double lam;
    double q;
    bool kkt;
    int next;  // pattern an active-set step would try next (-1: none)
};

// symmetric 3x3 matrix, upper triangle only (the QP matrix of the SQP: six registers instead of nine per lane)
struct Sym3 {
    double a00, a01, a02, a11, a12, a22;
    R2S_DEV double operator()(int i, int j) const
    {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        return lo == 0 ? (hi == 0 ? a00 : (hi == 1 ? a01 : a02)) : (lo == 1 ? (hi == 1 ? a11 : a12) : a22);
    }
};

// returns 0: pattern unusable, 2: primal infeasible, 1: primal feasible (o.kkt tells optimality)
R2S_DEV int qp_pattern(int pat, const Sym3& H, const double g[3], const double a[3], double e,
                       const double lo[3], const double hi[3], QpOut& o)
{
    o.next = -1;
    const int s[3] = {pat & 3, (pat >> 2) & 3, pat >> 4};   // two bits per variable: 0 free, 1 at the lower, 2 at the upper bound
    double dB[3], aa[3], b[3], M[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double l = lo[i], h = hi[i];   // loaded before the select: keeps the caller's state in registers
        dB[i] = (s[i] == 1) ? l : ((s[i] == 2) ? h : 0.0);
    }
    double ep = e;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (s[i]) {
            ep = fma(-a[i], dB[i], ep);
            aa[i] = 0.0;
            b[i] = 0.0;
        } else {
            aa[i] = a[i];
            double t = -g[i];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (s[j]) t = fma(-H(i, j), dB[j], t);
            b[i] = t;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i][j] = (s[i] || s[j]) ? ((i == j) ? 1.0 : 0.0) : H(i, j);
    }
    // inverse of the masked symmetric matrix by its adjugate (one division); the leading minors double as
    // the positive-definiteness test (Sylvester)
    const double M00 = M[0][0], M01 = M[0][1], M02 = M[0][2], M11 = M[1][1], M12 = M[1][2], M22 = M[2][2];
    const double c00 = fma(M11, M22, -(M12 * M12));
    const double c01 = fma(M02, M12, -(M01 * M22));
    const double c02 = fma(M01, M12, -(M02 * M11));
    const double c11 = fma(M00, M22, -(M02 * M02));
    const double c12 = fma(M01, M02, -(M00 * M12));
    const double c22 = fma(M00, M11, -(M01 * M01));
    const double det = dot3(M00, M01, M02, c00, c01, c02);
    if (!(M00 > 0.0 && c22 > 0.0 && det > 0.0)) return 0;
    const double rdet = 1.0 / det;
    double u[3], v[3];
    u[0] = dot3(c00, c01, c02, aa[0], aa[1], aa[2]) * rdet;
    u[1] = dot3(c01, c11, c12, aa[0], aa[1], aa[2]) * rdet;
    u[2] = dot3(c02, c12, c22, aa[0], aa[1], aa[2]) * rdet;
    v[0] = dot3(c00, c01, c02, b[0], b[1], b[2]) * rdet;
    v[1] = dot3(c01, c11, c12, b[0], b[1], b[2]) * rdet;
    v[2] = dot3(c02, c12, c22, b[0], b[1], b[2]) * rdet;
    double den = dot3(aa[0], aa[1], aa[2], u[0], u[1], u[2]);
    // vacuous equality: the constraint gradient has no component along the free variables (rho is constant on this
    // face of the element); the face problem is unconstrained if the fixed variables meet the equality (see the oracle)
    const bool vac = (aa[0] == 0.0 && aa[1] == 0.0 && aa[2] == 0.0);
    if (vac ? (ep != 0.0) : !(den > 0.0)) return 0;
    double lam = vac ? 0.0 : (dot3(aa[0], aa[1], aa[2], v[0], v[1], v[2]) - ep) / den;
    bool ok = true;
    double worst = 0.0;
    const int pw[3] = {1, 4, 16};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (s[i]) {
            o.d[i] = dB[i];
        } else {
            o.d[i] = fma(-lam, u[i], v[i]);
            if (!(o.d[i] >= lo[i] - 1e-12 && o.d[i] <= hi[i] + 1e-12)) {
                ok = false;
                const double below = (lo[i] - 1e-12) - o.d[i], above = o.d[i] - (hi[i] + 1e-12);
                const double viol = fmax(below, above);
                if (viol > worst) { worst = viol; o.next = pat + ((above > below) ? 2 : 1) * pw[i]; }
            }
        }
    }
    if (!ok) {
        // two variables fixed already: fixing the violated third one leaves no freedom for the equality,
        // so the walk restarts from the violated bound alone
        const int np = o.next;
        if (np >= 0 && (np & 3) && ((np >> 2) & 3) && (np >> 4)) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (!s[i]) o.next = np & (3 << (2 * i));
        }
        return 2;
    }
    double Hd[3], q = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Hd[i] = dot3(H(i, 0), H(i, 1), H(i, 2), o.d[0], o.d[1], o.d[2]);
        q = fma(o.d[i], fma(0.5, Hd[i], g[i]), q);
    }
    bool kkt = true;
    worst = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (s[i] && !(vac && a[i] != 0.0)) {
            const double z = fma(lam, a[i], Hd[i] + g[i]);
            const double viol = (s[i] == 1) ? -z : z;
            if (s[i] == 1 && !(z >= 0.0)) kkt = false;
            if (s[i] == 2 && !(z <= 0.0)) kkt = false;
            if (viol > worst) { worst = viol; o.next = pat - s[i] * pw[i]; }
        }
    }
    o.lam = lam;
    o.q = q;
    o.kkt = kkt;
    return 1;
}

// the oracle's search order {0, 1, 2, 3, 6, 9, 18, 4, 5, 7, 8, 10, 11, 19, 20, 12, 15, 21, 24} (base-3 digits there) in the
// device's pattern code: two bits per variable (decoding a digit is an AND instead of a division by 3 / 9)
static __constant__ int c_pat_order[19] = {0, 1, 2, 4, 8, 16, 32, 5, 6, 9, 10, 17, 18, 33, 34, 20, 24, 36, 40};

R2S_DEV bool spd3(double h00, double h01, double h02, double h11, double h12, double h22)
{
    // Sylvester's criterion on the symmetric matrix: leading minors, no division
    const double c00 = fma(h11, h22, -(h12 * h12));
    const double c01 = fma(h02, h12, -(h01 * h22));
    const double c02 = fma(h01, h12, -(h02 * h11));
    const double m2 = fma(h00, h11, -(h01 * h01));
    const double det = dot3(h00, h01, h02, c00, c01, c02);
    return (h00 > 0.0) && (m2 > 0.0) && (det > 0.0);
}

template <class ER>   // any record with C[8][3], Cr[8] (ElemRec in SGPRs, IsoElemLds per lane)
R2S_DEV void iso_eval_fc(const ER& E, const double x[3], double rt, const double xi[3], double& f,
                         double& c)
{
    double r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = x[i] - tri_eval_value(R2S_CX(E, i), xi);
    const double rho = tri_eval_value(R2S_CR(E), xi);
    f = dot3(r[0], r[1], r[2], r[0], r[1], r[2]);
    c = rho - rt;
}

#define R2S_QP_WALK 8       // active-set steps before the exhaustive search
#define R2S_ISO_MAXIT 100   // = the oracle's ISO_MAXIT
#define R2S_ISO_MAX_NONCONVEX 48   // = the oracle's ISO_MAX_NONCONVEX
#define R2S_ISO_TOL 1e-6    // = the oracle's ISO_TOL (see the note there)
#define R2S_ISO_MAX_RESTORE 3

// qp_pattern for the patterns with TWO / ONE fixed variable(s) without the masked 3x3 machinery: the same IEEE operations
// minus the ones whose operands the mask makes 0 or 1 (x * 1, x + 0 and 0 * finite are exact), so d, lam and q come out
// bit for bit as from qp_pattern (checked against the oracle's qp_pattern on 7.2 M random pattern solves incl.
// indefinite matrices, vanishing gradient components and degenerate bounds).  The non-convex search of
// iso_project_full evaluates all 19 patterns in every such iteration; these cost a fifth of the generic form.
// Return value as qp_pattern; kkt / next are not produced (the search takes the feasible pattern of least value).
template <int k /* free */>
R2S_DEV int qp_fixed2_t(int si, int sj, const Sym3& H, const double g[3], const double a[3], double e,
                        const double lo[3], const double hi[3], QpOut& o)
{
    constexpr int i = (k == 0) ? 1 : 0, j = (k == 2) ? 1 : 2;   // the fixed ones, ascending (compile-time indices: no
                                                                // array of the caller is indexed dynamically = no scratch)
    const double dBi = (si == 1) ? lo[i] : hi[i], dBj = (sj == 1) ? lo[j] : hi[j];
    double ep = e;
    ep = fma(-a[i], dBi, ep);
    ep = fma(-a[j], dBj, ep);
    double t = -g[k];
    t = fma(-H(k, i), dBi, t);
    t = fma(-H(k, j), dBj, t);
    const double Hkk = H(k, k);
    if (!(Hkk > 0.0)) return 0;
    const double rdet = 1.0 / Hkk;
    const double u = a[k] * rdet, v = t * rdet;
    const double den = a[k] * u;
    const bool vac = (a[k] == 0.0);
    if (vac ? (ep != 0.0) : !(den > 0.0)) return 0;
    const double lam = vac ? 0.0 : (a[k] * v - ep) / den;
    const double dk = fma(-lam, u, v);
    if (!(dk >= lo[k] - 1e-12 && dk <= hi[k] + 1e-12)) return 2;
    o.d[i] = dBi; o.d[j] = dBj; o.d[k] = dk;
    double q = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double Hd = dot3(H(r, 0), H(r, 1), H(r, 2), o.d[0], o.d[1], o.d[2]);
        q = fma(o.d[r], fma(0.5, Hd, g[r]), q);
    }
    o.lam = lam;
    o.q = q;
    return 1;
}

R2S_DEV int qp_fixed2(int k, int si, int sj, const Sym3& H, const double g[3], const double a[3], double e,
                      const double lo[3], const double hi[3], QpOut& o)
{
    if (k == 0) return qp_fixed2_t<0>(si, sj, H, g, a, e, lo, hi, o);
    if (k == 1) return qp_fixed2_t<1>(si, sj, H, g, a, e, lo, hi, o);
    return qp_fixed2_t<2>(si, sj, H, g, a, e, lo, hi, o);
}

template <int i /* fixed */>
R2S_DEV int qp_fixed1_t(int si, const Sym3& H, const double g[3], const double a[3], double e,
                        const double lo[3], const double hi[3], QpOut& o)
{
    constexpr int j = (i == 0) ? 1 : 0, k = (i == 2) ? 1 : 2;   // the free ones, ascending
    const double dBi = (si == 1) ? lo[i] : hi[i];
    const double ep = fma(-a[i], dBi, e);
    const double bj = fma(-H(j, i), dBi, -g[j]), bk = fma(-H(k, i), dBi, -g[k]);
    const double Hjj = H(j, j), Hjk = H(j, k), Hkk = H(k, k);
    // determinant and Sylvester minors of the masked matrix as qp_pattern rounds them (they differ with the position of
    // the fixed variable: the cofactor of the unit diagonal entry is fma(Hjj, Hkk, -(Hjk Hjk)), the expansion along the
    // first row of the other two cases is fma(Hjk, -Hjk, Hjj Hkk))
    const double c2 = fma(Hjj, Hkk, -(Hjk * Hjk));
    const double det = (i == 0) ? c2 : fma(Hjk, -Hjk, Hjj * Hkk);
    const double m00 = (i == 0) ? 1.0 : Hjj;
    const double c22 = (i == 2) ? c2 : Hjj;
    if (!(m00 > 0.0 && c22 > 0.0 && det > 0.0)) return 0;
    const double rdet = 1.0 / det;
    const double uj = fma(-Hjk, a[k], Hkk * a[j]) * rdet;
    const double uk = fma(Hjj, a[k], -Hjk * a[j]) * rdet;
    const double vj = fma(-Hjk, bk, Hkk * bj) * rdet;
    const double vk = fma(Hjj, bk, -Hjk * bj) * rdet;
    const double den = fma(a[k], uk, a[j] * uj);
    const bool vac = (a[j] == 0.0 && a[k] == 0.0);
    if (vac ? (ep != 0.0) : !(den > 0.0)) return 0;
    const double lam = vac ? 0.0 : (fma(a[k], vk, a[j] * vj) - ep) / den;
    const double dj = fma(-lam, uj, vj), dk = fma(-lam, uk, vk);
    if (!(dj >= lo[j] - 1e-12 && dj <= hi[j] + 1e-12)) return 2;
    if (!(dk >= lo[k] - 1e-12 && dk <= hi[k] + 1e-12)) return 2;
    o.d[i] = dBi; o.d[j] = dj; o.d[k] = dk;
    double q = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double Hd = dot3(H(r, 0), H(r, 1), H(r, 2), o.d[0], o.d[1], o.d[2]);
        q = fma(o.d[r], fma(0.5, Hd, g[r]), q);
    }
    o.lam = lam;
    o.q = q;
    return 1;
}

R2S_DEV int qp_fixed1(int i, int si, const Sym3& H, const double g[3], const double a[3], double e,
                      const double lo[3], const double hi[3], QpOut& o)
{
    if (i == 0) return qp_fixed1_t<0>(si, H, g, a, e, lo, hi, o);
    if (i == 1) return qp_fixed1_t<1>(si, H, g, a, e, lo, hi, o);
    return qp_fixed1_t<2>(si, H, g, a, e, lo, hi, o);
}

// H = G + S + sg a a^T (upper triangle), g' = g - sg e a: the QP data of the oracle's iso_project_hex8
R2S_DEV void iso_qp_data(const double G[3][3], const double S[3], const double a[3], const double g[3], double sg,
                         double e, Sym3& H, double gp[3])
{
    const double se = sg * e;
    const double sa0 = sg * a[0], sa1 = sg * a[1], sa2 = sg * a[2];
    H.a00 = fma(sa0, a[0], G[0][0]); H.a01 = fma(sa0, a[1], G[0][1]); H.a02 = fma(sa0, a[2], G[0][2]);
    H.a11 = fma(sa1, a[1], G[1][1]); H.a12 = fma(sa1, a[2], G[1][2]);
    H.a22 = fma(sa2, a[2], G[2][2]);
    H.a01 += S[0];
    H.a02 += S[1];
    H.a12 += S[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = fma(-se, a[i], g[i]);
}

// positive definite on the face of the box named by the pattern
R2S_DEV bool iso_face_spd(const Sym3& H, int pat)
{
    const bool f0 = (pat & 3) != 0, f1 = ((pat >> 2) & 3) != 0, f2 = (pat >> 4) != 0;
    return spd3(f0 ? 1.0 : H.a00, (f0 || f1) ? 0.0 : H.a01, (f0 || f2) ? 0.0 : H.a02,
                f1 ? 1.0 : H.a11, (f1 || f2) ? 0.0 : H.a12, f2 ? 1.0 : H.a22);
}

// warm-start pattern: a variable the last QP fixed stays fixed only if it sits on the element's own bound now
R2S_DEV int iso_clean_pattern(int pat, const double xi[3])
{
    const int s0 = pat & 3, s1 = (pat >> 2) & 3, s2 = pat >> 4;
    int p = 0;
    if ((s0 == 1 && xi[0] == -1.0) || (s0 == 2 && xi[0] == 1.0)) p += s0;
    if ((s1 == 1 && xi[1] == -1.0) || (s1 == 2 && xi[1] == 1.0)) p += 4 * s1;
    if ((s2 == 1 && xi[2] == -1.0) || (s2 == 2 && xi[2] == 1.0)) p += 16 * s2;
    return p;
}

// ---- the complete solver, one lane = one (element, voxel) pair ------------------------------------------
// Operation for operation the oracle's iso_project_hex8 (oracle/r2s_oracle.c): exact Lagrangian Hessian, convex
// active-set QP or the global minimiser of the non-convex QP over the trust region, L1 merit with second-order
// correction, restoration along nodal segments from stalls.  This straight-line form is the REFERENCE for the two lane
// machines below (the complete one of iso_straggler_kernel, phase by phase the same operations; the fast path of
// iso_project_hex_pl_kernel, its common case) and what iso_sweep_kernel runs when the straggler list overflowed.
template <class ER>
R2S_DEV bool iso_restore(const ER& E, const double x[3], double rt, const double xi[3], double c, double out[3])
{
    double bestf = INFINITY;
    bool found = false;
    for (int k = 0; k < 8; ++k) {
        const double nd[3] = {(k & 1) ? 1.0 : -1.0, (k & 2) ? 1.0 : -1.0, (k & 4) ? 1.0 : -1.0};
        const double ck = tri_eval_value(R2S_CR(E), nd) - rt;
        if ((c < 0.0) ? !(ck >= 0.0) : !(ck <= 0.0)) continue;
        const double dir[3] = {nd[0] - xi[0], nd[1] - xi[1], nd[2] - xi[2]};
        double tl = 0.0, th = 1.0, t = 1.0, p[3] = {nd[0], nd[1], nd[2]};
        if (ck != 0.0) {
            t = 0.5;
            for (int n = 0; n < 100; ++n) {
#pragma unroll
                for (int i = 0; i < 3; ++i) p[i] = fma(t, dir[i], xi[i]);
                const TriEval tr = tri_eval_full(R2S_CR(E), p);
                const double ph = tr.v - rt;
                if (ph == 0.0) break;
                if ((ph < 0.0) == (c < 0.0)) tl = t; else th = t;
                if (!(th - tl > 1e-15)) break;
                const double dph = dot3(tr.d1, tr.d2, tr.d3, dir[0], dir[1], dir[2]);
                double tn = t - ph / dph;
                if (!(tn > tl && tn < th)) tn = 0.5 * (tl + th);
                if (tn == t) break;
                t = tn;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) p[i] = fmin(fmax(fma(t, dir[i], xi[i]), -1.0), 1.0);
        }
        double fk, ckk;
        iso_eval_fc(E, x, rt, p, fk, ckk);
        if (fk < bestf) { bestf = fk; found = true; out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; }
    }
    return found;
}

// Returns the number of the last iteration (R2S_ISO_MAXIT + 1: cap).  Start state (xi, mu0, Delta0, pat0, it0): (0, 0, 2, 0, 0) for a run from the start, or the state at the beginning of
// the iteration in which the fast lane machine gave up - the same thing, provided no earlier iterate was exactly
// feasible (the run remembers the nearest feasible iterate for the case that it fails: IsoLane::seen).
template <class ER>
R2S_DEV int iso_project_full(const ER& E, double rmax_abs, const double x[3], double rt, double xi[3], double mu0 = 0.0,
                              double Delta0 = 2.0, int pat0 = 0, int it0 = 0)
{
    double mu = mu0, lam = 0.0, Delta = Delta0;
    int pat = pat0, nrest = 0;
    const double rtol = fmax(fabs(rt), rmax_abs) * 1e-14;
    double fbest = INFINITY, xbest[3] = {0.0, 0.0, 0.0};
    double sx[3] = {0.0, 0.0, 0.0}, smu = -1.0, sDelta = -1.0;   // cycle detection (see the oracle): the state at 16, 32, 64, 128
    int spat = -1, n_nonconvex = 0;   // (a hand-over has seen no non-convex model: the fast path gives up at the first)
    for (int it = it0; it < R2S_ISO_MAXIT; ++it) {
        double r[3], J[3][3], a[3], g[3], G[3][3], M2[3][3];
        if (it > 16 && xi[0] == sx[0] && xi[1] == sx[1] && xi[2] == sx[2] && mu == smu && Delta == sDelta && pat == spat) {
            if (fbest < INFINITY) { xi[0] = xbest[0]; xi[1] = xbest[1]; xi[2] = xbest[2]; }
            return R2S_ISO_MAXIT + 1;
        }
        if (it == 16 || it == 32 || it == 64 || it == 128) {   // (a hand-over starts at it0 <= R2S_ISO_FAST_IT = 16: no snapshot is missed)
            sx[0] = xi[0]; sx[1] = xi[1]; sx[2] = xi[2]; smu = mu; sDelta = Delta; spat = pat;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const TriEval t = tri_eval_full(R2S_CX(E, i), xi);
            r[i] = x[i] - t.v;
            J[i][0] = t.d1; J[i][1] = t.d2; J[i][2] = t.d3;
            M2[i][0] = t.m12; M2[i][1] = t.m13; M2[i][2] = t.m23;
        }
        const double f = dot3(r[0], r[1], r[2], r[0], r[1], r[2]);
        const TriEval tr = tri_eval_full(R2S_CR(E), xi);
        double c = tr.v - rt;
        a[0] = tr.d1; a[1] = tr.d2; a[2] = tr.d3;
        if (fabs(c) <= rtol) c = 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (fabs(a[j]) <= rtol) a[j] = 0.0;
        if (c == 0.0 && f < fbest) { fbest = f; xbest[0] = xi[0]; xbest[1] = xi[1]; xbest[2] = xi[2]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) g[j] = -2.0 * dot3(r[0], r[1], r[2], J[0][j], J[1][j], J[2][j]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i; j < 3; ++j)
                G[i][j] = G[j][i] = 2.0 * dot3(J[0][i], J[1][i], J[2][i], J[0][j], J[1][j], J[2][j]);
        pat = iso_clean_pattern(pat, xi);
        {
            double num = 0.0, den = 0.0;
            const int sp[3] = {pat & 3, (pat >> 2) & 3, pat >> 4};
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (!sp[i]) { num = fma(a[i], g[i], num); den = fma(a[i], a[i], den); }
            lam = (den > 0.0) ? -num / den : 0.0;
        }
        double S[3];
        {
            const double mr[3] = {tr.m12, tr.m13, tr.m23};
#pragma unroll
            for (int q = 0; q < 3; ++q)
                S[q] = fma(lam, mr[q], -2.0 * dot3(r[0], r[1], r[2], M2[0][q], M2[1][q], M2[2][q]));
        }
        double lo[3], hi[3], d[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lo[i] = fmax(-1.0 - xi[i], -Delta);
            hi[i] = fmin(1.0 - xi[i], Delta);
        }
        const double e = -c;
        double mplus = 0.0, mminus = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double p = a[i] * lo[i], q = a[i] * hi[i];
            mplus += fmax(p, q);
            mminus += fmin(p, q);
        }
        const double trG = G[0][0] + G[1][1] + G[2][2];
        const double aa2 = dot3(a[0], a[1], a[2], a[0], a[1], a[2]);
        const double sigma = 100.0 * trG / aa2;
        bool convex = true, corner = false, stall = false;
        int stop = 0;
        const bool near_feas = (fabs(c) <= 1e4 * rtol);
        const double mu_keep = (fabs(c) <= 1e10 * rtol) ? 0.5 : 1.0;
        double lam_new = lam, alpha = 1.0, qstep = 0.0, dGd = 0.0;
        if (e > mplus) {
#pragma unroll
            for (int i = 0; i < 3; ++i) d[i] = (a[i] > 0.0) ? hi[i] : ((a[i] < 0.0) ? lo[i] : 0.0);
            corner = true;
        } else if (e < mminus) {
#pragma unroll
            for (int i = 0; i < 3; ++i) d[i] = (a[i] > 0.0) ? lo[i] : ((a[i] < 0.0) ? hi[i] : 0.0);
            corner = true;
        }
        if (corner) {
            double bp = 0.0, bm = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double p = a[i] * (-1.0 - xi[i]), q = a[i] * (1.0 - xi[i]);
                bp += fmax(p, q);
                bm += fmin(p, q);
            }
            if ((e > 0.0) ? !(bp > 0.05 * e) : !(bm < 0.05 * e)) stall = true;
            double Gd[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) Gd[i] = dot3(G[i][0], G[i][1], G[i][2], d[0], d[1], d[2]);
            dGd = dot3(d[0], d[1], d[2], Gd[0], Gd[1], Gd[2]);
        } else {
            Sym3 H;
            double gp[3];
            int stage = 0;
            double sg = sigma;
            for (;;) {
                iso_qp_data(G, S, a, g, sg, e, H, gp);
                convex = iso_face_spd(H, pat);
                if (convex || stage) break;
                stage = 1;
                sg = 100.0 * sigma;
            }
            QpOut o;
            bool found = false;
            if (convex) {
                int p = pat;
                for (int step = 0; step < R2S_QP_WALK && p >= 0; ++step) {
                    const int rc = qp_pattern(p, H, gp, a, e, lo, hi, o);
                    if (rc == 0 && !stage) {
                        stage = 1;
                        sg = 100.0 * sigma;
                        iso_qp_data(G, S, a, g, sg, e, H, gp);
                        p = pat;
                        step = -1;
                        continue;
                    }
                    if (rc == 0) { convex = false; break; }
                    if (rc == 1 && o.kkt) {
                        found = true;
                        pat = p;
                        d[0] = o.d[0]; d[1] = o.d[1]; d[2] = o.d[2];
                        lam_new = o.lam;
                        qstep = o.q;
                        break;
                    }
                    p = o.next;
                }
            }
            if (!found) {
                double bestq = INFINITY;
                for (int ip = 0; ip < 19; ++ip) {
                    const int p = c_pat_order[ip];
                    int rc;
                    if (convex || ip == 0) {
                        rc = qp_pattern(p, H, gp, a, e, lo, hi, o);
                    } else if (ip < 7) {   // one variable fixed
                        const int s0 = p & 3, s1 = (p >> 2) & 3, s2 = p >> 4;
                        const int i = s0 ? 0 : (s1 ? 1 : 2);
                        rc = qp_fixed1(i, s0 + s1 + s2, H, gp, a, e, lo, hi, o);
                    } else {               // two fixed
                        const int s0 = p & 3, s1 = (p >> 2) & 3, s2 = p >> 4;
                        const int k = !s0 ? 0 : (!s1 ? 1 : 2);
                        rc = qp_fixed2(k, (k == 0) ? s1 : s0, (k == 2) ? s1 : s2, H, gp, a, e, lo, hi, o);
                    }
                    if (rc == 1) {
                        if ((convex && o.kkt) || o.q < bestq) {
                            bestq = o.q;
                            found = true;
                            pat = p;
                            d[0] = o.d[0]; d[1] = o.d[1]; d[2] = o.d[2];
                            lam_new = o.lam;
                            qstep = o.q;
                        }
                        if (convex && o.kkt) break;
                    }
                }
            }
            if (!found) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    d[i] = (e > 0.0) ? ((a[i] > 0.0) ? hi[i] : ((a[i] < 0.0) ? lo[i] : 0.0))
                                     : ((a[i] > 0.0) ? lo[i] : ((a[i] < 0.0) ? hi[i] : 0.0));
                corner = true;
                convex = true;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) d[i] = fmin(fmax(d[i], lo[i]), hi[i]);
        }
        if (!convex && ++n_nonconvex > R2S_ISO_MAX_NONCONVEX) {
            if (fbest < INFINITY) { xi[0] = xbest[0]; xi[1] = xbest[1]; xi[2] = xbest[2]; }
            return R2S_ISO_MAXIT + 1;
        }
        const double dmax = fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
        const double ad = dot3(a[0], a[1], a[2], d[0], d[1], d[2]);
        const double pred_c = fabs(c) - fabs(c + ad);
        if (stall) {
            stop = 2;
        } else if (!convex) {
            double mu_t = fmax(mu_keep * mu, 2.0 * fabs(lam_new));
            double pred = fma(mu_t, pred_c, -qstep);
            if (!(pred > 0.0)) {
                if (pred_c > 0.0) { mu_t = 2.0 * qstep / pred_c; pred = qstep; }
                else stop = near_feas ? 3 : 2;
            }
            if (!stop && !(pred > 1e-14 * f)) stop = 3;
            if (!stop) {
                mu = mu_t;
                double xt[3], ft, ct;
#pragma unroll
                for (int i = 0; i < 3; ++i) xt[i] = fmin(fmax(xi[i] + d[i], -1.0), 1.0);
                iso_eval_fc(E, x, rt, xt, ft, ct);
                const double phi0 = fma(mu, fabs(c), f);
                if (!(phi0 - fma(mu, fabs(ct), ft) >= 1e-4 * pred)) {
                    const int sp[3] = {pat & 3, (pat >> 2) & 3, pat >> 4};
                    double den = 0.0, d2[3] = {d[0], d[1], d[2]};
                    bool ok = false;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (!sp[i]) den = fma(a[i], a[i], den);
                    if (den > 0.0) {
                        const double sc = -ct / den;
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            if (!sp[i]) d2[i] = fma(sc, a[i], d[i]);
                            xt[i] = fmin(fmax(xi[i] + d2[i], -1.0), 1.0);
                        }
                        double f2, c2;
                        iso_eval_fc(E, x, rt, xt, f2, c2);
                        ok = (phi0 - fma(mu, fabs(c2), f2) >= 1e-4 * pred);
                    }
                    if (ok) { d[0] = d2[0]; d[1] = d2[1]; d[2] = d2[2]; }
                    else alpha = 0.0;
                }
            }
        } else if (!(dmax > R2S_ISO_TOL)) {
            stop = (corner && !near_feas) ? 2 : 1;
        } else {
            const double gd = dot3(g[0], g[1], g[2], d[0], d[1], d[2]);
            double mu_t = corner ? mu : fmax(mu_keep * mu, 2.0 * fabs(lam_new));
            if (corner && pred_c > 0.0) {
                const double need = 2.0 * fma(0.5, dGd, gd) / pred_c;
                if (need > mu_t) mu_t = need;
            }
            if (!(fma(-mu_t, pred_c, gd) < 0.0)) {
                if (pred_c > 0.0) mu_t = 2.0 * gd / pred_c;
                else stop = (near_feas && !corner) ? 4 : 2;
            }
            if (!stop) {
                mu = mu_t;
                const double D = fma(-mu, pred_c, gd);
                const double phi0 = fma(mu, fabs(c), f);
                for (int ls = 0; ls < 30; ++ls) {
                    double xt[3], ft, ct;
#pragma unroll
                    for (int i = 0; i < 3; ++i) xt[i] = fmin(fmax(fma(alpha, d[i], xi[i]), -1.0), 1.0);
                    iso_eval_fc(E, x, rt, xt, ft, ct);
                    if (fma(mu, fabs(ct), ft) <= fma(1e-4 * alpha, D, phi0)) break;
                    if (ls == 0 && !corner) {
                        const int sp[3] = {pat & 3, (pat >> 2) & 3, pat >> 4};
                        double den = 0.0;
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            if (!sp[i]) den = fma(a[i], a[i], den);
                        if (den > 0.0) {
                            const double sc = -ct / den;
                            double d2[3];
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                d2[i] = sp[i] ? d[i] : fma(sc, a[i], d[i]);
                                xt[i] = fmin(fmax(xi[i] + d2[i], -1.0), 1.0);
                            }
                            double f2, c2;
                            iso_eval_fc(E, x, rt, xt, f2, c2);
                            if (fma(mu, fabs(c2), f2) <= fma(1e-4, D, phi0)) {
                                d[0] = d2[0]; d[1] = d2[1]; d[2] = d2[2];
                                break;
                            }
                        }
                    }
                    alpha *= 0.5;
                }
            }
        }
        Delta = (alpha < 1.0) ? ((alpha > 0.0) ? alpha * dmax : 0.25 * dmax) : fmin(2.0, fmax(Delta, 2.0 * dmax));
        if (stop == 2) {
            double xr[3];
            if (!near_feas && nrest < R2S_ISO_MAX_RESTORE && iso_restore(E, x, rt, xi, c, xr)) {
                xi[0] = xr[0]; xi[1] = xr[1]; xi[2] = xr[2];
                mu = 0.0; Delta = 2.0; pat = 0; nrest++;
                continue;
            }
            if (fbest < INFINITY) { xi[0] = xbest[0]; xi[1] = xbest[1]; xi[2] = xbest[2]; }
            return R2S_ISO_MAXIT + 1;   // (a stall no restoration repairs: no KKT point - the oracle's return value, counted by the callers)
        }
        if (stop == 3 || stop == 4) return it + 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) xi[i] = fmin(fmax(fma(alpha, d[i], xi[i]), -1.0), 1.0);
        if (stop == 1) return it + 1;
    }
    if (fbest < INFINITY) { xi[0] = xbest[0]; xi[1] = xbest[1]; xi[2] = xbest[2]; }
    return R2S_ISO_MAXIT + 1;
}

// ---- the complete solver as a per-lane state machine -----------------------------------------------------
// iso_straggler_kernel: the handed-over pairs are worked off by persistent wavefronts with lane refill, like the fast
// path, but every rule of iso_project_full is there - as phases, so that a trip costs what its lanes' phases cost (a
// plain iteration: one trip of ~900 instructions) instead of the union of all branches of the straight-line solver
// (~2 800 per wavefront-iteration).  Each phase performs exactly the IEEE operations of the corresponding part of
// iso_project_full, in the same order.
//   EVAL  cycle test, fields, QP data (first sigma, then the second), corner / stall test   -> QP | ENUM | POST
//   QP    ONE active-set pattern of the walk per visit                                       -> QP | EVAL (second sigma) | ENUM | POST
//   ENUM  the whole 19-pattern search (convex: first KKT pattern; non-convex: least value)  -> POST
//   POST  caps, step test, merit parameter; the non-convex trial with its correction        -> LS | UPD
//   LS    ONE back-tracking trial per visit (the first with the second-order correction)    -> LS | UPD
//   UPD   trust region, restoration, iterate update                                          -> EVAL | DONE
enum { FS_IDLE = 0, FS_EVAL, FS_QP, FS_ENUM, FS_POST, FS_LS, FS_UPD, FS_DONE };

struct IsoFullLane {
    double x[3];
    double xi[3], mu, Delta;
    double fbest, xbest[3];
    double sx[3], smu, sDelta;
    Sym3 H;
    double a[3], g[3], d[3];
    double se;       // sg * e of the QP data, or the curvature d.G.d of a corner step
    double f, c, lam_new, qstep, dmax;
    double alpha, D, phi0;
    int pat, it, nrest, n_nc, spat;
    int p, step, ls, phase, stop;
    bool corner, convex, stall, stage2, force2;
};

R2S_DEV void iso_full_start(IsoFullLane& s, const double x[3], const double xi[3], double mu, double Delta, int pat, int it)
{
    s.x[0] = x[0]; s.x[1] = x[1]; s.x[2] = x[2];
    s.xi[0] = xi[0]; s.xi[1] = xi[1]; s.xi[2] = xi[2];
    s.mu = mu; s.Delta = Delta; s.pat = pat; s.it = it;
    s.nrest = 0; s.n_nc = 0;
    s.fbest = INFINITY; s.xbest[0] = s.xbest[1] = s.xbest[2] = 0.0;
    s.sx[0] = s.sx[1] = s.sx[2] = 0.0; s.smu = -1.0; s.sDelta = -1.0; s.spat = -1;
    s.force2 = false;
    s.phase = FS_EVAL;
}

R2S_DEV void iso_full_fail(IsoFullLane& s)
{
    if (s.fbest < INFINITY) { s.xi[0] = s.xbest[0]; s.xi[1] = s.xbest[1]; s.xi[2] = s.xbest[2]; }
    s.it = R2S_ISO_MAXIT + 1;   // (what iso_project_full returns for a run that ends without a KKT point: counted, r2s_stats::n_iso_fail)
    s.phase = FS_DONE;
}

template <class ER>
R2S_DEV void iso_full_eval(const ER& E, double rt, double rtol, IsoFullLane& s)
{
    if (s.it > 16 && s.xi[0] == s.sx[0] && s.xi[1] == s.sx[1] && s.xi[2] == s.sx[2] && s.mu == s.smu && s.Delta == s.sDelta &&
        s.pat == s.spat && !s.force2) {
        iso_full_fail(s);
        return;
    }
    if ((s.it == 16 || s.it == 32 || s.it == 64 || s.it == 128) && !s.force2) {
        s.sx[0] = s.xi[0]; s.sx[1] = s.xi[1]; s.sx[2] = s.xi[2]; s.smu = s.mu; s.sDelta = s.Delta; s.spat = s.pat;
    }
    double r[3], J[3][3], G[3][3], M2[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const TriEval t = tri_eval_full(R2S_CX(E, i), s.xi);
        r[i] = s.x[i] - t.v;
        J[i][0] = t.d1; J[i][1] = t.d2; J[i][2] = t.d3;
        M2[i][0] = t.m12; M2[i][1] = t.m13; M2[i][2] = t.m23;
    }
    const double f = dot3(r[0], r[1], r[2], r[0], r[1], r[2]);
    const TriEval tr = tri_eval_full(R2S_CR(E), s.xi);
    double c = tr.v - rt;
    s.a[0] = tr.d1; s.a[1] = tr.d2; s.a[2] = tr.d3;
    if (fabs(c) <= rtol) c = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (fabs(s.a[j]) <= rtol) s.a[j] = 0.0;
    if (c == 0.0 && f < s.fbest) { s.fbest = f; s.xbest[0] = s.xi[0]; s.xbest[1] = s.xi[1]; s.xbest[2] = s.xi[2]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) s.g[j] = -2.0 * dot3(r[0], r[1], r[2], J[0][j], J[1][j], J[2][j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j)
            G[i][j] = G[j][i] = 2.0 * dot3(J[0][i], J[1][i], J[2][i], J[0][j], J[1][j], J[2][j]);
    s.pat = iso_clean_pattern(s.pat, s.xi);
    double lam;
    {
        double num = 0.0, den = 0.0;
        const int sp[3] = {s.pat & 3, (s.pat >> 2) & 3, s.pat >> 4};
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (!sp[i]) { num = fma(s.a[i], s.g[i], num); den = fma(s.a[i], s.a[i], den); }
        lam = (den > 0.0) ? -num / den : 0.0;
    }
    double S[3];
    {
        const double mr[3] = {tr.m12, tr.m13, tr.m23};
#pragma unroll
        for (int q = 0; q < 3; ++q)
            S[q] = fma(lam, mr[q], -2.0 * dot3(r[0], r[1], r[2], M2[0][q], M2[1][q], M2[2][q]));
    }
    double lo[3], hi[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        lo[i] = fmax(-1.0 - s.xi[i], -s.Delta);
        hi[i] = fmin(1.0 - s.xi[i], s.Delta);
    }
    const double e = -c;
    double mplus = 0.0, mminus = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double p = s.a[i] * lo[i], q = s.a[i] * hi[i];
        mplus += fmax(p, q);
        mminus += fmin(p, q);
    }
    const double trG = G[0][0] + G[1][1] + G[2][2];
    const double aa2 = dot3(s.a[0], s.a[1], s.a[2], s.a[0], s.a[1], s.a[2]);
    const double sigma = 100.0 * trG / aa2;
    s.convex = true; s.corner = false; s.stall = false; s.stop = 0;
    s.lam_new = lam; s.alpha = 1.0; s.qstep = 0.0;
    s.f = f; s.c = c;
    if (e > mplus) {
#pragma unroll
        for (int i = 0; i < 3; ++i) s.d[i] = (s.a[i] > 0.0) ? hi[i] : ((s.a[i] < 0.0) ? lo[i] : 0.0);
        s.corner = true;
    } else if (e < mminus) {
#pragma unroll
        for (int i = 0; i < 3; ++i) s.d[i] = (s.a[i] > 0.0) ? lo[i] : ((s.a[i] < 0.0) ? hi[i] : 0.0);
        s.corner = true;
    }
    if (s.corner) {
        double bp = 0.0, bm = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double p = s.a[i] * (-1.0 - s.xi[i]), q = s.a[i] * (1.0 - s.xi[i]);
            bp += fmax(p, q);
            bm += fmin(p, q);
        }
        if ((e > 0.0) ? !(bp > 0.05 * e) : !(bm < 0.05 * e)) s.stall = true;
        double Gd[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) Gd[i] = dot3(G[i][0], G[i][1], G[i][2], s.d[0], s.d[1], s.d[2]);
        s.se = dot3(s.d[0], s.d[1], s.d[2], Gd[0], Gd[1], Gd[2]);
        s.phase = FS_POST;
    } else {
        double gp[3];
        int stage = s.force2 ? 1 : 0;
        double sg = stage ? 100.0 * sigma : sigma;
        for (;;) {
            iso_qp_data(G, S, s.a, s.g, sg, e, s.H, gp);
            // (force2: the walk of this iteration ran into a face on which the first matrix is not positive definite - the
            //  complete solver rebuilds the matrix with the second sigma and restarts the walk WITHOUT testing the warm-start
            //  face again; the walk's first pattern performs the same Sylvester test, so testing it here changes nothing)
            s.convex = iso_face_spd(s.H, s.pat);
            if (s.convex || stage) break;
            stage = 1;
            sg = 100.0 * sigma;
        }
        s.se = sg * e;
        s.stage2 = (stage != 0);
        s.p = s.pat;
        s.step = 0;
        s.phase = s.convex ? FS_QP : FS_ENUM;
    }
    s.force2 = false;
}

R2S_DEV void iso_full_bounds(const IsoFullLane& s, double lo[3], double hi[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        lo[i] = fmax(-1.0 - s.xi[i], -s.Delta);
        hi[i] = fmin(1.0 - s.xi[i], s.Delta);
    }
}

R2S_DEV void iso_full_qp(IsoFullLane& s)
{
    QpOut o;
    double lo[3], hi[3], gp[3];
    iso_full_bounds(s, lo, hi);
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = fma(-s.se, s.a[i], s.g[i]);
    const double e = -s.c;
    const int rc = qp_pattern(s.p, s.H, gp, s.a, e, lo, hi, o);
    if (rc == 0 && !s.stage2) {
        s.force2 = true;          // once more with the larger sigma, from the warm-start pattern
        s.phase = FS_EVAL;
    } else if (rc == 0) {
        s.convex = false;
        s.phase = FS_ENUM;
    } else if (rc == 1 && o.kkt) {
        s.pat = s.p;
#pragma unroll
        for (int i = 0; i < 3; ++i) s.d[i] = fmin(fmax(o.d[i], lo[i]), hi[i]);
        s.lam_new = o.lam;
        s.qstep = o.q;
        s.phase = FS_POST;
    } else {
        s.p = o.next;
        s.step += 1;
        if (!(s.step < R2S_QP_WALK && s.p >= 0)) s.phase = FS_ENUM;   // exhaustive search (convex flavour)
    }
}

R2S_DEV void iso_full_enum(IsoFullLane& s)
{
    QpOut o;
    double lo[3], hi[3], gp[3];
    iso_full_bounds(s, lo, hi);
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = fma(-s.se, s.a[i], s.g[i]);
    const double e = -s.c;
    bool found = false;
    double bestq = INFINITY;
    for (int ip = 0; ip < 19; ++ip) {
        const int p = c_pat_order[ip];
        int rc;
        if (s.convex || ip == 0) {
            rc = qp_pattern(p, s.H, gp, s.a, e, lo, hi, o);
        } else if (ip < 7) {   // one variable fixed
            const int s0 = p & 3, s1 = (p >> 2) & 3, s2 = p >> 4;
            const int i = s0 ? 0 : (s1 ? 1 : 2);
            rc = qp_fixed1(i, s0 + s1 + s2, s.H, gp, s.a, e, lo, hi, o);
        } else {               // two fixed
            const int s0 = p & 3, s1 = (p >> 2) & 3, s2 = p >> 4;
            const int k = !s0 ? 0 : (!s1 ? 1 : 2);
            rc = qp_fixed2(k, (k == 0) ? s1 : s0, (k == 2) ? s1 : s2, s.H, gp, s.a, e, lo, hi, o);
        }
        if (rc == 1) {
            if ((s.convex && o.kkt) || o.q < bestq) {
                bestq = o.q;
                found = true;
                s.pat = p;
                s.d[0] = o.d[0]; s.d[1] = o.d[1]; s.d[2] = o.d[2];
                s.lam_new = o.lam;
                s.qstep = o.q;
            }
            if (s.convex && o.kkt) break;
        }
    }
    if (!found) {   // numerically degenerate: corner move towards feasibility (without the curvature term of the penalty rule)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            s.d[i] = (e > 0.0) ? ((s.a[i] > 0.0) ? hi[i] : ((s.a[i] < 0.0) ? lo[i] : 0.0))
                               : ((s.a[i] > 0.0) ? lo[i] : ((s.a[i] < 0.0) ? hi[i] : 0.0));
        s.corner = true;
        s.convex = true;
        s.se = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) s.d[i] = fmin(fmax(s.d[i], lo[i]), hi[i]);
    s.phase = FS_POST;
}

// iso_full_enum for ONE lane `src` (wave-uniform) by the whole wavefront: lane ip < 19 solves pattern ip of the owner's
// QP, the winner is picked by the rule of the loop above (first pattern in c_pat_order with a KKT point when the model is
// convex, else the first one with the smallest model value).  Same numbers as the loop - qp_pattern and its specialised
// forms agree bit for bit - at the latency of one pattern instead of nineteen: the straggler kernel lasts as long as its
// slowest pairs, and those are the ones whose iterations end up here.  All 64 lanes call it; only lane `src` changes.
R2S_DEV void iso_full_enum_coop(IsoFullLane& s, const int src)
{
    double lo[3], hi[3], gp[3];
    iso_full_bounds(s, lo, hi);
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = fma(-s.se, s.a[i], s.g[i]);
    auto bcast = [src](double v) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
    };
    Sym3 H;
    H.a00 = bcast(s.H.a00); H.a01 = bcast(s.H.a01); H.a02 = bcast(s.H.a02);
    H.a11 = bcast(s.H.a11); H.a12 = bcast(s.H.a12); H.a22 = bcast(s.H.a22);
    double a[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { a[i] = bcast(s.a[i]); gp[i] = bcast(gp[i]); lo[i] = bcast(lo[i]); hi[i] = bcast(hi[i]); }
    const double e = bcast(-s.c);
    const bool convex = __builtin_amdgcn_readlane((int)s.convex, src) != 0;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int pat = c_pat_order[lane < 19 ? lane : 0];
    QpOut o;
    const int rc = qp_pattern(pat, H, gp, a, e, lo, hi, o);
    const bool ok = lane < 19 && rc == 1;
    int win = -1;
    const unsigned long long mk = __ballot(ok && o.kkt);
    if (convex && mk) {
        win = __builtin_ctzll(mk);
    } else {
        double q = (ok && o.q < INFINITY) ? o.q : INFINITY;   // (the loop takes a pattern only when q < best so far)
        double qmin = q;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {   // lanes 0..18 hold the candidates: five butterfly steps reach them all
            const double t = __hiloint2double(__shfl_xor(__double2hiint(qmin), off), __shfl_xor(__double2loint(qmin), off));
            qmin = fmin(qmin, t);
        }
        const unsigned long long mq = __ballot(q < INFINITY && q == qmin);
        if (mq) win = __builtin_ctzll(mq);
    }
    double wd[3] = {0.0, 0.0, 0.0}, wlam = 0.0, wq = 0.0;
    if (win >= 0) {
        auto from = [win](double v) {
            return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), win), __builtin_amdgcn_readlane(__double2loint(v), win));
        };
        wd[0] = from(o.d[0]); wd[1] = from(o.d[1]); wd[2] = from(o.d[2]);
        wlam = from(o.lam);
        wq = from(o.q);
    }
    if (lane != src) return;
    iso_full_bounds(s, lo, hi);
    const double es = -s.c;
    if (win >= 0) {
        s.pat = c_pat_order[win];
        s.d[0] = wd[0]; s.d[1] = wd[1]; s.d[2] = wd[2];
        s.lam_new = wlam;
        s.qstep = wq;
    } else {   // numerically degenerate: corner move towards feasibility (as in iso_full_enum)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            s.d[i] = (es > 0.0) ? ((s.a[i] > 0.0) ? hi[i] : ((s.a[i] < 0.0) ? lo[i] : 0.0))
                                : ((s.a[i] > 0.0) ? lo[i] : ((s.a[i] < 0.0) ? hi[i] : 0.0));
        s.corner = true;
        s.convex = true;
        s.se = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) s.d[i] = fmin(fmax(s.d[i], lo[i]), hi[i]);
    s.phase = FS_POST;
}

template <class ER>
R2S_DEV void iso_full_post(const ER& E, double rt, double rtol, IsoFullLane& s)
{
    if (!s.convex && ++s.n_nc > R2S_ISO_MAX_NONCONVEX) { iso_full_fail(s); return; }
    const double dmax = fmax(fabs(s.d[0]), fmax(fabs(s.d[1]), fabs(s.d[2])));
    s.dmax = dmax;
    const double ad = dot3(s.a[0], s.a[1], s.a[2], s.d[0], s.d[1], s.d[2]);
    const double pred_c = fabs(s.c) - fabs(s.c + ad);
    const bool near_feas = (fabs(s.c) <= 1e4 * rtol);
    const double mu_keep = (fabs(s.c) <= 1e10 * rtol) ? 0.5 : 1.0;
    s.phase = FS_UPD;
    if (s.stall) {
        s.stop = 2;
    } else if (!s.convex) {
        double mu_t = fmax(mu_keep * s.mu, 2.0 * fabs(s.lam_new));
        double pred = fma(mu_t, pred_c, -s.qstep);
        if (!(pred > 0.0)) {
            if (pred_c > 0.0) { mu_t = 2.0 * s.qstep / pred_c; pred = s.qstep; }
            else s.stop = near_feas ? 3 : 2;
        }
        if (!s.stop && !(pred > 1e-14 * s.f)) s.stop = 3;
        if (!s.stop) {
            s.mu = mu_t;
            double xt[3], ft, ct;
#pragma unroll
            for (int i = 0; i < 3; ++i) xt[i] = fmin(fmax(s.xi[i] + s.d[i], -1.0), 1.0);
            iso_eval_fc(E, s.x, rt, xt, ft, ct);
            const double phi0 = fma(s.mu, fabs(s.c), s.f);
            if (!(phi0 - fma(s.mu, fabs(ct), ft) >= 1e-4 * pred)) {
                const int sp[3] = {s.pat & 3, (s.pat >> 2) & 3, s.pat >> 4};
                double den = 0.0, d2[3] = {s.d[0], s.d[1], s.d[2]};
                bool ok = false;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (!sp[i]) den = fma(s.a[i], s.a[i], den);
                if (den > 0.0) {
                    const double sc = -ct / den;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        if (!sp[i]) d2[i] = fma(sc, s.a[i], s.d[i]);
                        xt[i] = fmin(fmax(s.xi[i] + d2[i], -1.0), 1.0);
                    }
                    double f2, c2;
                    iso_eval_fc(E, s.x, rt, xt, f2, c2);
                    ok = (phi0 - fma(s.mu, fabs(c2), f2) >= 1e-4 * pred);
                }
                if (ok) { s.d[0] = d2[0]; s.d[1] = d2[1]; s.d[2] = d2[2]; }
                else s.alpha = 0.0;
            }
        }
    } else if (!(dmax > R2S_ISO_TOL)) {
        s.stop = (s.corner && !near_feas) ? 2 : 1;
    } else {
        const double gd = dot3(s.g[0], s.g[1], s.g[2], s.d[0], s.d[1], s.d[2]);
        double mu_t = s.corner ? s.mu : fmax(mu_keep * s.mu, 2.0 * fabs(s.lam_new));
        if (s.corner && pred_c > 0.0) {
            const double need = 2.0 * fma(0.5, s.se, gd) / pred_c;
            if (need > mu_t) mu_t = need;
        }
        if (!(fma(-mu_t, pred_c, gd) < 0.0)) {
            if (pred_c > 0.0) mu_t = 2.0 * gd / pred_c;
            else s.stop = (near_feas && !s.corner) ? 4 : 2;
        }
        if (!s.stop) {
            s.mu = mu_t;
            s.D = fma(-s.mu, pred_c, gd);
            s.phi0 = fma(s.mu, fabs(s.c), s.f);
            s.ls = 0;
            s.phase = FS_LS;
        }
    }
}

template <class ER>
R2S_DEV void iso_full_ls(const ER& E, double rt, IsoFullLane& s)
{
    double xt[3], ft, ct;
#pragma unroll
    for (int i = 0; i < 3; ++i) xt[i] = fmin(fmax(fma(s.alpha, s.d[i], s.xi[i]), -1.0), 1.0);
    iso_eval_fc(E, s.x, rt, xt, ft, ct);
    if (fma(s.mu, fabs(ct), ft) <= fma(1e-4 * s.alpha, s.D, s.phi0)) { s.phase = FS_UPD; return; }
    if (s.ls == 0 && !s.corner) {
        const int sp[3] = {s.pat & 3, (s.pat >> 2) & 3, s.pat >> 4};
        double den = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (!sp[i]) den = fma(s.a[i], s.a[i], den);
        if (den > 0.0) {
            const double sc = -ct / den;
            double d2[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                d2[i] = sp[i] ? s.d[i] : fma(sc, s.a[i], s.d[i]);
                xt[i] = fmin(fmax(s.xi[i] + d2[i], -1.0), 1.0);
            }
            double f2, c2;
            iso_eval_fc(E, s.x, rt, xt, f2, c2);
            if (fma(s.mu, fabs(c2), f2) <= fma(1e-4, s.D, s.phi0)) {
                s.d[0] = d2[0]; s.d[1] = d2[1]; s.d[2] = d2[2];
                s.phase = FS_UPD;
                return;
            }
        }
    }
    s.alpha *= 0.5;
    s.ls += 1;
    if (s.ls == 30) s.phase = FS_UPD;
}

template <class ER>
R2S_DEV void iso_full_upd(const ER& E, double rt, double rtol, IsoFullLane& s)
{
    s.Delta = (s.alpha < 1.0) ? ((s.alpha > 0.0) ? s.alpha * s.dmax : 0.25 * s.dmax) : fmin(2.0, fmax(s.Delta, 2.0 * s.dmax));
    if (s.stop == 2) {
        const bool near_feas = (fabs(s.c) <= 1e4 * rtol);
        double xr[3];
        if (!near_feas && s.nrest < R2S_ISO_MAX_RESTORE && iso_restore(E, s.x, rt, s.xi, s.c, xr)) {
            s.xi[0] = xr[0]; s.xi[1] = xr[1]; s.xi[2] = xr[2];
            s.mu = 0.0; s.Delta = 2.0; s.pat = 0; s.nrest++;
            s.it += 1;
            if (s.it == R2S_ISO_MAXIT) iso_full_fail(s); else s.phase = FS_EVAL;
            return;
        }
        iso_full_fail(s);
        return;
    }
    if (s.stop == 3 || s.stop == 4) { s.phase = FS_DONE; return; }
#pragma unroll
    for (int i = 0; i < 3; ++i) s.xi[i] = fmin(fmax(fma(s.alpha, s.d[i], s.xi[i]), -1.0), 1.0);
    if (s.stop == 1) { s.phase = FS_DONE; return; }
    s.it += 1;
    if (s.it == R2S_ISO_MAXIT) iso_full_fail(s); else s.phase = FS_EVAL;
}

// ---- the fast path of the same solver as a per-lane state machine ---------------------------------------
// iso_project_hex_pl_kernel runs the COMMON case of the iteration above with the lanes of a wavefront in
// different phases / iterations / voxels: plain SQP steps - the QP convex on the faces the active-set walk visits
// (first sigma), found within R2S_QP_WALK patterns, the full step accepted by the merit function at the first trial -
// and feasibility steps into a corner of the trust region.  Every phase performs exactly the IEEE operations of the
// corresponding part of iso_project_full, in the same order.  Anything else (non-convex model, second sigma, exhaustive
// pattern search, rejected step, stall, more than R2S_ISO_FAST_IT iterations) ends in ISO_BAIL: the pair goes to the
// straggler list with its state at the start of the iteration (xi, mu, Delta, pattern, iteration count) and the complete
// lane machine continues from there (from xi = 0 when an earlier iterate was exactly feasible: IsoLane::seen).  The fast
// path therefore never has to agree with the complete solver beyond the point where it bails - and needs neither
// line-search nor pattern-search state.
//   EVAL    fields, QP data, corner test            -> QP | FINISH | BAIL
//   QP      ONE active-set pattern per visit        -> QP | FINISH | BAIL
//   FINISH  step test, merit parameter, the full-step trial, trust region, iterate update -> EVAL | DONE | BAIL
#ifndef R2S_ISO_FAST_IT
#define R2S_ISO_FAST_IT 16
#endif
enum { ISO_IDLE = 0, ISO_EVAL, ISO_QP, ISO_FINISH, ISO_DONE, ISO_BAIL };

struct IsoLane {
    double x[3];
    double xi[3], mu, Delta;
    Sym3 H;
    double a[3], g[3], d[3];
    double se;   // sigma * e (QP steps) or the curvature d.G.d of a corner step
    double f, c, lam_new;
    int pat, it, p, step, phase;
    bool corner;
    bool seen;   // an EARLIER iterate of this run was exactly feasible (c == 0): a hand-over has to start from xi = 0 then
};

// step bounds of the current iterate: the box |xi| <= 1 cut with the trust region
R2S_DEV void iso_lane_bounds(const IsoLane& s, double lo[3], double hi[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        lo[i] = fmax(-1.0 - s.xi[i], -s.Delta);
        hi[i] = fmin(1.0 - s.xi[i], s.Delta);
    }
}

R2S_DEV void iso_lane_start(IsoLane& s, const double x[3])
{
    s.x[0] = x[0]; s.x[1] = x[1]; s.x[2] = x[2];
    s.xi[0] = s.xi[1] = s.xi[2] = 0.0;
    s.mu = 0.0; s.Delta = 2.0;
    s.pat = 0; s.it = 0;
    s.seen = false;
    s.phase = ISO_EVAL;
}

template <class ER>
R2S_DEV void iso_lane_eval(const ER& E, double rt, double rtol, IsoLane& s)
{
    double r[3], J[3][3], G[3][3], M2[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const TriEval t = tri_eval_full(R2S_CX(E, i), s.xi);
        r[i] = s.x[i] - t.v;
        J[i][0] = t.d1; J[i][1] = t.d2; J[i][2] = t.d3;
        M2[i][0] = t.m12; M2[i][1] = t.m13; M2[i][2] = t.m23;
    }
    const double f = dot3(r[0], r[1], r[2], r[0], r[1], r[2]);
    const TriEval tr = tri_eval_full(R2S_CR(E), s.xi);
    double c = tr.v - rt;
    s.a[0] = tr.d1; s.a[1] = tr.d2; s.a[2] = tr.d3;
    if (fabs(c) <= rtol) c = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (fabs(s.a[j]) <= rtol) s.a[j] = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) s.g[j] = -2.0 * dot3(r[0], r[1], r[2], J[0][j], J[1][j], J[2][j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j)
            G[i][j] = G[j][i] = 2.0 * dot3(J[0][i], J[1][i], J[2][i], J[0][j], J[1][j], J[2][j]);
    s.pat = iso_clean_pattern(s.pat, s.xi);
    double lam;
    {
        double num = 0.0, den = 0.0;
        const int sp[3] = {s.pat & 3, (s.pat >> 2) & 3, s.pat >> 4};
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (!sp[i]) { num = fma(s.a[i], s.g[i], num); den = fma(s.a[i], s.a[i], den); }
        lam = (den > 0.0) ? -num / den : 0.0;
    }
    double S[3];
    {
        const double mr[3] = {tr.m12, tr.m13, tr.m23};
#pragma unroll
        for (int q = 0; q < 3; ++q)
            S[q] = fma(lam, mr[q], -2.0 * dot3(r[0], r[1], r[2], M2[0][q], M2[1][q], M2[2][q]));
    }
    double lo[3], hi[3];
    iso_lane_bounds(s, lo, hi);
    const double e = -c;
    double mplus = 0.0, mminus = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double p = s.a[i] * lo[i], q = s.a[i] * hi[i];
        mplus += fmax(p, q);
        mminus += fmin(p, q);
    }
    const double trG = G[0][0] + G[1][1] + G[2][2];
    const double aa2 = dot3(s.a[0], s.a[1], s.a[2], s.a[0], s.a[1], s.a[2]);
    const double sigma = 100.0 * trG / aa2;
    s.corner = false;
    s.lam_new = lam;
    s.f = f; s.c = c;
    if (e > mplus) {
#pragma unroll
        for (int i = 0; i < 3; ++i) s.d[i] = (s.a[i] > 0.0) ? hi[i] : ((s.a[i] < 0.0) ? lo[i] : 0.0);
        s.corner = true;
    } else if (e < mminus) {
#pragma unroll
        for (int i = 0; i < 3; ++i) s.d[i] = (s.a[i] > 0.0) ? lo[i] : ((s.a[i] < 0.0) ? hi[i] : 0.0);
        s.corner = true;
    }
    if (s.corner) {
        // how much of the violation the linear model can remove anywhere in the element (see the oracle)
        double bp = 0.0, bm = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double p = s.a[i] * (-1.0 - s.xi[i]), q = s.a[i] * (1.0 - s.xi[i]);
            bp += fmax(p, q);
            bm += fmin(p, q);
        }
        const bool stall = (e > 0.0) ? !(bp > 0.05 * e) : !(bm < 0.05 * e);
        double Gd[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) Gd[i] = dot3(G[i][0], G[i][1], G[i][2], s.d[0], s.d[1], s.d[2]);
        s.se = dot3(s.d[0], s.d[1], s.d[2], Gd[0], Gd[1], Gd[2]);
        s.phase = stall ? ISO_BAIL : ISO_FINISH;
    } else {
        double gp[3];
        iso_qp_data(G, S, s.a, s.g, sigma, e, s.H, gp);
        s.se = sigma * e;
        s.p = s.pat;
        s.step = 0;
        s.phase = iso_face_spd(s.H, s.pat) ? ISO_QP : ISO_BAIL;
    }
}

R2S_DEV void iso_lane_qp(IsoLane& s)
{
    QpOut o;
    double lo[3], hi[3], gp[3];
    iso_lane_bounds(s, lo, hi);
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = fma(-s.se, s.a[i], s.g[i]);
    const double e = -s.c;
    const int rc = qp_pattern(s.p, s.H, gp, s.a, e, lo, hi, o);
    if (rc == 1 && o.kkt) {
        // active-set walk: the first primal feasible KKT pattern is the minimiser (s.p; it becomes the warm-start
        // pattern when the step is accepted - until then xi, mu, Delta, pat are the state at the iteration's start,
        // which is what a hand-over passes on)
#pragma unroll
        for (int i = 0; i < 3; ++i) s.d[i] = fmin(fmax(o.d[i], lo[i]), hi[i]);
        s.lam_new = o.lam;
        s.phase = ISO_FINISH;
    } else if (rc == 0) {
        s.phase = ISO_BAIL;   // second sigma / non-convex model: the complete solver's business
    } else {
        s.p = o.next;
        s.step += 1;
        if (!(s.step < R2S_QP_WALK && s.p >= 0)) s.phase = ISO_BAIL;   // exhaustive search
    }
}

template <class ER>
R2S_DEV void iso_lane_finish(const ER& E, double rt, double rtol, IsoLane& s)
{
    const double dmax = fmax(fabs(s.d[0]), fmax(fabs(s.d[1]), fabs(s.d[2])));
    const bool near_feas = (fabs(s.c) <= 1e4 * rtol);
    if (!(dmax > R2S_ISO_TOL)) {
        if (s.corner && !near_feas) { s.phase = ISO_BAIL; return; }   // stuck off the iso-surface: restoration
    } else {
        const double ad = dot3(s.a[0], s.a[1], s.a[2], s.d[0], s.d[1], s.d[2]);
        const double pred_c = fabs(s.c) - fabs(s.c + ad);
        const double gd = dot3(s.g[0], s.g[1], s.g[2], s.d[0], s.d[1], s.d[2]);
        const double mu_keep = (fabs(s.c) <= 1e10 * rtol) ? 0.5 : 1.0;
        double mu_t = s.corner ? s.mu : fmax(mu_keep * s.mu, 2.0 * fabs(s.lam_new));
        if (s.corner && pred_c > 0.0) {
            // a step that only buys feasibility: the merit function pays for the growth of f including its curvature
            const double need = 2.0 * fma(0.5, s.se, gd) / pred_c;
            if (need > mu_t) mu_t = need;
        }
        if (!(fma(-mu_t, pred_c, gd) < 0.0)) {
            if (pred_c > 0.0) mu_t = 2.0 * gd / pred_c;
            else {
                // no descent on the merit function: from a feasible point the QP step is rounding noise (converged,
                // the iterate stays); otherwise restoration
                s.phase = (near_feas && !s.corner) ? ISO_DONE : ISO_BAIL;
                return;
            }
        }
        const double D = fma(-mu_t, pred_c, gd);
        const double phi0 = fma(mu_t, fabs(s.c), s.f);
        double xt[3], ft, ct;
#pragma unroll
        for (int i = 0; i < 3; ++i) xt[i] = fmin(fmax(fma(1.0, s.d[i], s.xi[i]), -1.0), 1.0);
        iso_eval_fc(E, s.x, rt, xt, ft, ct);
        if (!(fma(mu_t, fabs(ct), ft) <= fma(1e-4 * 1.0, D, phi0))) { s.phase = ISO_BAIL; return; }   // correction / back-tracking
        s.mu = mu_t;
        s.Delta = fmin(2.0, fmax(s.Delta, 2.0 * dmax));
        if (!s.corner) s.pat = s.p;
        s.seen = s.seen || (s.c == 0.0);
#pragma unroll
        for (int i = 0; i < 3; ++i) s.xi[i] = xt[i];
        s.it += 1;
        s.phase = (s.it == R2S_ISO_FAST_IT) ? ISO_BAIL : ISO_EVAL;
        return;
    }
    // converged: the last (tiny) step is applied
    s.Delta = fmin(2.0, fmax(s.Delta, 2.0 * dmax));
#pragma unroll
    for (int i = 0; i < 3; ++i) s.xi[i] = fmin(fmax(fma(1.0, s.d[i], s.xi[i]), -1.0), 1.0);
    s.phase = ISO_DONE;
}

// running minimum of one voxel: WriteValue / update_distance_parallel!
// (sdfOnDensityField.jl:44-57, :121-136) - strict '<' on the magnitude.
struct VoxState {
    double cur;    // |dist_local[v]|, starts at 1e10 (sdfOnDensityField.jl:172-183)
    double xp[3];  // projection point, starts at 0 (:174)
};

// sym (order-independent mode): an exact tie goes to the lexicographically smaller projection point
R2S_DEV bool write_value(VoxState& s, double d, const double xp[3], bool sym = false)
{
    if (sym && fabs(d) == s.cur) {
        const bool less = xp[0] < s.xp[0] || (xp[0] == s.xp[0] && (xp[1] < s.xp[1] || (xp[1] == s.xp[1] && xp[2] < s.xp[2])));
        if (less) { s.xp[0] = xp[0]; s.xp[1] = xp[1]; s.xp[2] = xp[2]; }
        return false;
    }
    if (fabs(d) < s.cur) {
        s.cur = d;
        s.xp[0] = xp[0]; s.xp[1] = xp[1]; s.xp[2] = xp[2];
        return true;
    }
    return false;
}

// IsProjectedOnFullSegment, HEX8 (sdfOnDensityField.jl:78-119)
R2S_DEV bool projected_on_full_segment(VoxState& s, const ElemRec& E, double rt, const double xp[3],
                                       const double x[3], bool sym = false)
{
    double xi[3], N[8];
    inv_map_hex8(E, xp, xi);
    const double m = fmax(fabs(xi[0]), fmax(fabs(xi[1]), fabs(xi[2])));
    if (m < 1.001) {
        hex8_shape(xi, N);
        double rho = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) rho += N[k] * E.r[k];
        if (rho >= rt) {
            write_value(s, norm3(x[0] - xp[0], x[1] - xp[1], x[2] - xp[2]), xp, sym);
            return true;
        }
    }
    return false;
}

template <class Rec>
R2S_DEV bool tri_candidate(VoxState& s, bool solid, const Rec& E, double rt, const double xp[3],
                           const double x[3], double d, bool sym = false)
{
    return solid ? write_value(s, d, xp, sym) : projected_on_full_segment(s, E, rt, xp, x, sym);
}

// SURVEY 8(f)4, order-independent mode: every candidate of the triangle takes part in the minimum - the foot on
// the face (if inside), the feet on the three edges (if on the segment) and the three vertices
template <class Rec>
R2S_DEV void process_triangle_sym(VoxState& s, const BandItem& T, const Rec& E, double rt, const double x[3],
                                                  bool solid, bool inside, double l0, double l1, double l2)
{
    double xp[3];
    if (inside) {
#pragma unroll
        for (int i = 0; i < 3; ++i) xp[i] = l0 * T.tri[0][i] + l1 * T.tri[1][i] + l2 * T.tri[2][i];
        tri_candidate(s, solid, E, rt, xp, x, norm3(x[0] - xp[0], x[1] - xp[1], x[2] - xp[2]), true);
    }
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
        const double L = T.L[j];
        const double P = (x[0] - T.tri[j][0]) * T.eh[j][0] + (x[1] - T.tri[j][1]) * T.eh[j][1] +
                         (x[2] - T.tri[j][2]) * T.eh[j][2];
        if (P >= 0 && P <= L) {
#pragma unroll
            for (int i = 0; i < 3; ++i) xp[i] = T.tri[j][i] + T.eh[j][i] * P;
            tri_candidate(s, solid, E, rt, xp, x, norm3(x[0] - xp[0], x[1] - xp[1], x[2] - xp[2]), true);
        }
    }
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
        xp[0] = T.tri[j][0]; xp[1] = T.tri[j][1]; xp[2] = T.tri[j][2];
        tri_candidate(s, solid, E, rt, xp, x, norm3(x[0] - xp[0], x[1] - xp[1], x[2] - xp[2]), true);
    }
}

// process_triangle_projection! for one voxel (sdfOnDensityField.jl:675-813)
template <class Rec, bool SYM = false>
R2S_DEV void process_triangle(VoxState& s, const BandItem& T, const Rec& E, double rt,
                              const double x[3])
{
    const bool solid = (T.kind == 1);
    // barycentricCoordinates (TriangularMeshUtils.jl:1-24) with the factored matrix
    double b0 = x[1] * T.n[2] - x[2] * T.n[1];
    double b1 = x[2] * T.n[0] - x[0] * T.n[2];
    double b2 = x[0] * T.n[1] - x[1] * T.n[0];
    if (T.im == 0) b0 = 1.0; else if (T.im == 1) b1 = 1.0; else b2 = 1.0;
    // forward substitution with the stored row swaps (the factors are stored in their
    // final row order, so the second swap is applied before the first elimination step)
    if (T.p0 == 1) { double t = b0; b0 = b1; b1 = t; }
    else if (T.p0 == 2) { double t = b0; b0 = b2; b2 = t; }
    if (T.p1 == 2) { double t = b1; b1 = b2; b2 = t; }
    b1 -= T.lu[1][0] * b0;
    b2 -= T.lu[2][0] * b0;
    b2 -= T.lu[2][1] * b1;
    double l2 = b2 / T.lu[2][2];
    double l1 = (b1 - T.lu[1][2] * l2) / T.lu[1][1];
    double l0 = (b0 - T.lu[0][1] * l1 - T.lu[0][2] * l2) / T.lu[0][0];
    if (T.sing) { l0 = l1 = l2 = NAN; }
    bool ok = false;
    double xp[3];
    // minimum(lam) >= 0 with Julia's NaN-propagating minimum
    const bool inside = (l0 >= 0.0) && (l1 >= 0.0) && (l2 >= 0.0);
    if constexpr (SYM) {   // order-independent mode: its own instantiation, the reference path keeps its registers
        process_triangle_sym(s, T, E, rt, x, solid, inside, l0, l1, l2);
        return;
    }
    if (inside) {
#pragma unroll
        for (int i = 0; i < 3; ++i) xp[i] = l0 * T.tri[0][i] + l1 * T.tri[1][i] + l2 * T.tri[2][i];
        const double d = norm3(x[0] - xp[0], x[1] - xp[1], x[2] - xp[2]);
        ok = tri_candidate(s, solid, E, rt, xp, x, d);
    } else {
#pragma unroll 1
        for (int j = 0; j < 3; ++j) {
            const double L = T.L[j];
            const double P = (x[0] - T.tri[j][0]) * T.eh[j][0] + (x[1] - T.tri[j][1]) * T.eh[j][1] +
                             (x[2] - T.tri[j][2]) * T.eh[j][2];
            if (P >= 0 && P <= L) {
#pragma unroll
                for (int i = 0; i < 3; ++i) xp[i] = T.tri[j][i] + T.eh[j][i] * P;
                const double d = norm3(x[0] - xp[0], x[1] - xp[1], x[2] - xp[2]);
                ok = tri_candidate(s, solid, E, rt, xp, x, d);
                if (ok) break;
            }
        }
    }
    if (!ok) {
        double dd[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) dd[j] = norm3(x[0] - T.tri[j][0], x[1] - T.tri[j][1], x[2] - T.tri[j][2]);
        int idx = 0;
        double dm = dd[0];
        if (dd[1] < dm || (dd[1] != dd[1] && dm == dm)) { idx = 1; dm = dd[1]; }
        if (dd[2] < dm || (dd[2] != dd[2] && dm == dm)) { idx = 2; dm = dd[2]; }
        xp[0] = (idx == 0) ? T.tri[0][0] : ((idx == 1) ? T.tri[1][0] : T.tri[2][0]);
        xp[1] = (idx == 0) ? T.tri[0][1] : ((idx == 1) ? T.tri[1][1] : T.tri[2][1]);
        xp[2] = (idx == 0) ? T.tri[0][2] : ((idx == 1) ? T.tri[1][2] : T.tri[2][2]);
        tri_candidate(s, solid, E, rt, xp, x, dm);
    }
}


// =====================================================================================
// TET4
// =====================================================================================
// One pre-gathered TET4 element with the factorisations its per-voxel tests need.
// (three parts so that elem_prep_kernel can leave the middle one out for the 97 % of the elements that are no iso items)
struct alignas(16) TetGeom {
    double X[4][3];
    double r[4];
    double mn[3];
    double mx[3];
    double rmax;
    double rmin;
    double lu3[3][3];  // partial-pivot LU of [x2-x1, x3-x1, x4-x1]   (FindLocalCoordinates.jl:124-129)
    double lu4[4][4];  // partial-pivot LU of [v1 v2 v3 v4; 1 1 1 1]  (SignDetection.jl:236-239)
    int32_t p3[2];     // row swaps of lu3 (column c swapped with row p3[c])
    int32_t p4[3];     // row swaps of lu4
    int32_t sing3, sing4;
    int32_t blo[3], bhi[3];  // 1-based bin range of create_grid_tetrahedra_mapping_TET4 (:191-192)
    int32_t pad;
};
// per-element constants of the iso-surface projection (tet4_iso_constants): inverse of the edge matrix,
// density gradient and its square, and per face the segment {rho = rho_t} n face (start, direction, length^2).
// Only computed and stored for elements the iso-surface passes through (class ISO).
struct alignas(16) TetIso {
    double Ai[3][3];
    double gr[3];
    double g2;
    double sa[4][3];
    double sab[4][3];
    double sab2[4];
    int32_t segok[4];
};
// unit outward face normals fn[f] with two offsets each (tet4_face_planes): fn[f].x > fo[f] => the barycentric
// coordinate of face f is below -1e-6, so is_point_in_tetrahedron (tolerance 1e-10, SignDetection.jl:220-242)
// rejects the point; fn[f].x < fi[f] for all four faces => every coordinate is above +1e-6 and it accepts it.
// Only the points in between (a 2e-6 shell around the faces) need the 4x4 solve.  fo = +inf, fi = -inf for
// degenerate or flat elements (height / longest edge < 1e-3), which always take the solve.
struct alignas(16) TetPlanes {
    double fn[4][3];
    double fo[4];
    double fi[4];
};
struct alignas(16) TetRec : TetGeom, TetIso, TetPlanes {};

__device__ const int c_tet_isn[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {0, 3, 2}};

// shape_functions(TET4, l) (ShapeFunctions.jl:18-28)
R2S_DEV void tet4_shape(const double l[3], double N[4])
{
    N[0] = l[0]; N[1] = l[1]; N[2] = l[2];
    N[3] = 1.0 - ((l[0] + l[1]) + l[2]);
}

// find_local_coordinates, TET4 (FindLocalCoordinates.jl:110-149)
R2S_DEV bool find_local_tet4(const TetRec& E, const double x[3], double loc[3])
{
    double b0 = x[0] - E.X[0][0], b1 = x[1] - E.X[0][1], b2 = x[2] - E.X[0][2];
    if (E.p3[0] == 1) { double t = b0; b0 = b1; b1 = t; }
    else if (E.p3[0] == 2) { double t = b0; b0 = b2; b2 = t; }
    if (E.p3[1] == 2) { double t = b1; b1 = b2; b2 = t; }
    b1 -= E.lu3[1][0] * b0;
    b2 -= E.lu3[2][0] * b0;
    b2 -= E.lu3[2][1] * b1;
    const double l4 = b2 / E.lu3[2][2];
    const double l3 = (b1 - E.lu3[1][2] * l4) / E.lu3[1][1];
    const double l2 = (b0 - E.lu3[0][1] * l3 - E.lu3[0][2] * l4) / E.lu3[0][0];
    const double l1 = 1.0 - ((l2 + l3) + l4);
    // validate_local_coords(TET4, [l1,l2,l3,l4]) (ElementTypes.jl:104-106)
    const bool ok = !E.sing3 && l1 >= 0.0 && l2 >= 0.0 && l3 >= 0.0 && l4 >= 0.0 && (((l1 + l2) + l3) + l4) <= 1.0;
    if (!ok) { loc[0] = loc[1] = loc[2] = 10.0; return false; }
    loc[0] = l1; loc[1] = l2; loc[2] = l3;
    return true;
}

// IsProjectedOnFullSegment, TET4 branch (sdfOnDensityField.jl:92-113)
R2S_DEV bool projected_on_full_segment(VoxState& s, const TetRec& E, double rt, const double xp[3], const double x[3],
                                       bool sym = false)
{
    double loc[3], N[4];
    find_local_tet4(E, xp, loc);
    const double sum = (loc[0] + loc[1]) + loc[2];
    const bool valid = loc[0] >= 0.0 && loc[1] >= 0.0 && loc[2] >= 0.0 && sum <= 1.0 && sum <= 1.001;
    if (valid) {
        tet4_shape(loc, N);
        double rho = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) rho += N[k] * E.r[k];
        if (rho >= rt) {
            write_value(s, norm3(x[0] - xp[0], x[1] - xp[1], x[2] - xp[2]), xp, sym);
            return true;
        }
    }
    return false;
}

// compute_coords_on_iso, TET4 (ComputeCoordsOnIso.jl:90-181): closest point of the planar polygon
// {p in tet : rho(p) = rho_t}, closed form (see the oracle / DESIGN.md); natural coordinates out.
// Everything that does not depend on the voxel is evaluated once per element (tet4_iso_constants) with the
// same expressions, in the same order, as the oracle evaluates them per voxel.
R2S_DEV void tet4_iso_constants(TetRec& E, double rt)
{
    double A[3][3];
    for (int i = 0; i < 3; ++i) {
        A[i][0] = E.X[1][i] - E.X[0][i];
        A[i][1] = E.X[2][i] - E.X[0][i];
        A[i][2] = E.X[3][i] - E.X[0][i];
    }
    const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    const double c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    E.Ai[0][0] = c00 / det; E.Ai[1][0] = c01 / det; E.Ai[2][0] = c02 / det;
    E.Ai[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
    E.Ai[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
    E.Ai[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
    E.Ai[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
    E.Ai[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    E.Ai[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    const double dr[3] = {E.r[1] - E.r[0], E.r[2] - E.r[0], E.r[3] - E.r[0]};
    for (int i = 0; i < 3; ++i) E.gr[i] = E.Ai[0][i] * dr[0] + E.Ai[1][i] * dr[1] + E.Ai[2][i] * dr[2];
    E.g2 = E.gr[0] * E.gr[0] + E.gr[1] * E.gr[1] + E.gr[2] * E.gr[2];
    for (int f = 0; f < 4; ++f) {
        double P[3][3];
        int np = 0;
        for (int e = 0; e < 3; ++e) {
            const int a = c_tet_isn[f][e], b = c_tet_isn[f][(e + 1) % 3];
            const double ra = E.r[a] - rt, rb = E.r[b] - rt;
            if (ra * rb <= 0.0 && E.r[a] != E.r[b]) {
                const double t = (rt - E.r[a]) / (E.r[b] - E.r[a]);
                P[np][0] = E.X[a][0] + t * (E.X[b][0] - E.X[a][0]);
                P[np][1] = E.X[a][1] + t * (E.X[b][1] - E.X[a][1]);
                P[np][2] = E.X[a][2] + t * (E.X[b][2] - E.X[a][2]);
                np++;
            }
        }
        E.segok[f] = np >= 2;
        double a3[3] = {0, 0, 0}, b3[3] = {0, 0, 0};
        if (np >= 2) {
            for (int i = 0; i < 3; ++i) { a3[i] = P[0][i]; b3[i] = P[1][i]; }
            if (np == 3) {
                double d01 = 0, d02 = 0, d12 = 0;
                for (int i = 0; i < 3; ++i) {
                    d01 += (P[0][i] - P[1][i]) * (P[0][i] - P[1][i]);
                    d02 += (P[0][i] - P[2][i]) * (P[0][i] - P[2][i]);
                    d12 += (P[1][i] - P[2][i]) * (P[1][i] - P[2][i]);
                }
                if (d02 >= d01 && d02 >= d12) { for (int i = 0; i < 3; ++i) b3[i] = P[2][i]; }
                else if (d12 >= d01 && d12 >= d02) {
                    for (int i = 0; i < 3; ++i) { a3[i] = P[1][i]; b3[i] = P[2][i]; }
                }
            }
        }
        double ab2 = 0.0;
        for (int i = 0; i < 3; ++i) {
            E.sa[f][i] = a3[i];
            E.sab[f][i] = b3[i] - a3[i];
            ab2 += E.sab[f][i] * E.sab[f][i];
        }
        E.sab2[f] = ab2;
    }
}

R2S_DEV void iso_project_tet4(const TetRec& E, const double x[3], double rt, double lam[3])
{
    const double d0[3] = {x[0] - E.X[0][0], x[1] - E.X[0][1], x[2] - E.X[0][2]};
    const double rho_x = E.r[0] + (E.gr[0] * d0[0] + E.gr[1] * d0[1] + E.gr[2] * d0[2]);
    const double tq = (rho_x - rt) / E.g2;
    double best[3] = {0, 0, 0}, bestd = INFINITY;
    {
        const double q[3] = {x[0] - tq * E.gr[0], x[1] - tq * E.gr[1], x[2] - tq * E.gr[2]};
        const double dq[3] = {q[0] - E.X[0][0], q[1] - E.X[0][1], q[2] - E.X[0][2]};
        double l[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) l[i + 1] = E.Ai[i][0] * dq[0] + E.Ai[i][1] * dq[1] + E.Ai[i][2] * dq[2];
        l[0] = 1.0 - ((l[1] + l[2]) + l[3]);
        if (l[0] >= 0.0 && l[1] >= 0.0 && l[2] >= 0.0 && l[3] >= 0.0) {
            best[0] = q[0]; best[1] = q[1]; best[2] = q[2];
            bestd = 0.0;
        }
    }
    if (bestd != 0.0) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (E.segok[f]) {
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double ax = x[i] - E.sa[f][i];
                    dot += ax * E.sab[f][i];
                }
                double s = E.sab2[f] > 0.0 ? dot / E.sab2[f] : 0.0;
                s = fmin(fmax(s, 0.0), 1.0);
                double p[3], dd = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    p[i] = E.sa[f][i] + s * E.sab[f][i];
                    dd += (x[i] - p[i]) * (x[i] - p[i]);
                }
                if (dd < bestd) { bestd = dd; best[0] = p[0]; best[1] = p[1]; best[2] = p[2]; }
            }
        }
    }
    if (bestd == INFINITY) { lam[0] = lam[1] = lam[2] = 0.25; return; }
    const double db[3] = {best[0] - E.X[0][0], best[1] - E.X[0][1], best[2] - E.X[0][2]};
    double l234[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) l234[i] = E.Ai[i][0] * db[0] + E.Ai[i][1] * db[1] + E.Ai[i][2] * db[2];
    lam[0] = 1.0 - ((l234[0] + l234[1]) + l234[2]);
    lam[1] = l234[0];
    lam[2] = l234[1];
}

R2S_DEV double iso_candidate(const TetRec& E, double rt, const double x[3], double xp[3])
{
    double lam[3], N[4];
    iso_project_tet4(E, x, rt, lam);
    tet4_shape(lam, N);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) t += E.X[k][i] * N[k];
        xp[i] = t;
    }
    return norm3(x[0] - xp[0], x[1] - xp[1], x[2] - xp[2]);
}

// unit outward normals and the two offsets of the four faces (TetRec::fn, fo, fi)
R2S_DEV void tet4_face_planes(TetRec& E)
{
    double scale = 0.0, edge2 = 0.0;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 3; ++i) scale = fmax(scale, fabs(E.X[k][i]));
    for (int a = 0; a < 4; ++a)
        for (int b = a + 1; b < 4; ++b) {
            const double d[3] = {E.X[b][0] - E.X[a][0], E.X[b][1] - E.X[a][1], E.X[b][2] - E.X[a][2]};
            edge2 = fmax(edge2, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        }
    const double edge = sqrt(edge2);
    bool good = edge > 0.0 && !E.sing3 && !E.sing4;
    for (int f = 0; f < 4; ++f) {   // face f = the three vertices other than f
        const int a = (f + 1) & 3, b = (f + 2) & 3, c = (f + 3) & 3;
        const double u[3] = {E.X[b][0] - E.X[a][0], E.X[b][1] - E.X[a][1], E.X[b][2] - E.X[a][2]};
        const double v[3] = {E.X[c][0] - E.X[a][0], E.X[c][1] - E.X[a][1], E.X[c][2] - E.X[a][2]};
        double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
        const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(nn > 0.0)) { good = false; E.fn[f][0] = E.fn[f][1] = E.fn[f][2] = 0.0; continue; }
        n[0] /= nn; n[1] /= nn; n[2] /= nn;
        const double d = n[0] * E.X[a][0] + n[1] * E.X[a][1] + n[2] * E.X[a][2];
        double h = d - (n[0] * E.X[f][0] + n[1] * E.X[f][1] + n[2] * E.X[f][2]);   // height of vertex f over the face (signed)
        double ds = d;
        if (h < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; h = -h; ds = -d; }   // outward = away from vertex f
        if (!(h > 1e-3 * edge)) good = false;
        const double m = 1e-6 * h + 1e-11 * scale;   // 1e-6 of the coordinate + the rounding of fn.x itself
        E.fn[f][0] = n[0]; E.fn[f][1] = n[1]; E.fn[f][2] = n[2];
        E.fo[f] = ds + m;
        E.fi[f] = ds - m;
    }
    if (!good)
        for (int f = 0; f < 4; ++f) { E.fo[f] = INFINITY; E.fi[f] = -INFINITY; }
}
// position of p against the face planes: outside = beyond some face by the margin, inside = within all four by it
R2S_DEV void tet4_classify(const TetRec& E, const double p[3], bool& outside, bool& inside)
{
    outside = false;
    inside = true;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const double t = E.fn[f][0] * p[0] + E.fn[f][1] * p[1] + E.fn[f][2] * p[2];
        outside = outside || (t > E.fo[f]);
        inside = inside && (t < E.fi[f]);
    }
}

// is_point_in_tetrahedron (SignDetection.jl:220-242), tolerance 1e-10
R2S_DEV bool point_in_tet(const TetRec& E, const double p[3])
{
    const double tol = 1e-10;
    if (p[0] < E.mn[0] - tol || p[0] > E.mx[0] + tol || p[1] < E.mn[1] - tol || p[1] > E.mx[1] + tol ||
        p[2] < E.mn[2] - tol || p[2] > E.mx[2] + tol)
        return false;
    if (E.sing4) return false;
    double b0 = p[0], b1 = p[1], b2 = p[2], b3 = 1.0;
    // row swaps of the factorisation, in order
    if (E.p4[0] == 1) { double t = b0; b0 = b1; b1 = t; }
    else if (E.p4[0] == 2) { double t = b0; b0 = b2; b2 = t; }
    else if (E.p4[0] == 3) { double t = b0; b0 = b3; b3 = t; }
    if (E.p4[1] == 2) { double t = b1; b1 = b2; b2 = t; }
    else if (E.p4[1] == 3) { double t = b1; b1 = b3; b3 = t; }
    if (E.p4[2] == 3) { double t = b2; b2 = b3; b3 = t; }
    b1 -= E.lu4[1][0] * b0;
    b2 -= E.lu4[2][0] * b0;
    b3 -= E.lu4[3][0] * b0;
    b2 -= E.lu4[2][1] * b1;
    b3 -= E.lu4[3][1] * b1;
    b3 -= E.lu4[3][2] * b2;
    const double l3 = b3 / E.lu4[3][3];
    const double l2 = (b2 - E.lu4[2][3] * l3) / E.lu4[2][2];
    const double l1 = (b1 - E.lu4[1][2] * l2 - E.lu4[1][3] * l3) / E.lu4[1][1];
    const double l0 = (b0 - E.lu4[0][1] * l1 - E.lu4[0][2] * l2 - E.lu4[0][3] * l3) / E.lu4[0][0];
    return l0 >= -tol && l0 <= 1.0 + tol && l1 >= -tol && l1 <= 1.0 + tol && l2 >= -tol && l2 <= 1.0 + tol &&
           l3 >= -tol && l3 <= 1.0 + tol;
}

}  // namespace r2s
