// RBF stencil products as a row walk (included by r2s_post.hip after RbfLutVals / exp_neg_fast / rbf_tap_order).
//
//   MODE 0:  y = K x            (compute_sparse_kernel_matrix + the CG's product, RBFs4Smoothing.jl:142-176, 191-202)
//   MODE 1:  out = sum_j w_j exp(-(dist/sigma)^2) + add   with one target per lattice point
//                                (rbf_interpolation_kdtree, RBFs4Smoothing.jl:219-248)
//
// Both are (2R+1)^3-point stencils whose coefficients depend on the lattice position only through the three Float32
// coordinate differences (rbf_lut_axis: a handful of distinct values - "variants" - per axis and offset).  The
// coefficient of tap (dk, dj, di) at lattice point (i, j, k) is  T[dk][c(k,dk)][dj][b(j,dj)][di][a(i,di)].
//
// Data flow of one workgroup (NW wavefronts = 64 NW consecutive i, one plane k, L consecutive rows j):
//   * x: every lane keeps the (2R+1) x (2R+1) rows around its output row in REGISTERS (each with its 2R+1 shifted
//     copies) and walks along j: a step loads one new row per plane (2R+1 rows instead of (2R+1)^2), issued one step
//     ahead.  The loads are buffer loads - descriptor and row offset in SGPRs, the lane part a precomputed 32-bit
//     offset - so the VALU carries the products only.
//   * coefficients: per step the workgroup stages the rows T[dk][c][dj][b][:][:] of its (j, k) in LDS (contiguous
//     (2R+1) x NV chunks, double-buffered, fetched one step ahead, ONE barrier per step); a lane picks its entry with
//     one LDS read per tap (address = its x variant + an immediate).
//   * absent neighbours (outside the lattice, or beyond the kernel support) have coefficient 0 (variant slot NV-1 is
//     all zero, and T holds 0 where the reference's threshold test fails) and a clamped address: adding 0 * (finite
//     value) leaves the Float32 / Float64 partial sums unchanged, so the sums equal the reference's, which skips
//     those neighbours, bit for bit - with no predicates in the loop.
//   * order of the additions: MODE 0 ascending (dk, dj, di) = ascending column index of the sparse matrix; MODE 1 by
//     lattice distance^2, then (dz, dy, dx) (build_stencil), Float64 product and sum rounded to Float32 per neighbour.
#pragma once

#include <type_traits>

#define RBF_WALK_LMAX 32

struct RbfWalkArgs {
    int nx, ny, nz;
    const uint8_t *vx, *vy, *vz;   // [2R+1][n]: variant id of (offset, index), 255 = no such neighbour
    const void* T;                 // [dk][c][dj][b][di][a]: float (MODE 0) or double (MODE 1), NV slots per variant axis
    const float* x;                // input vector, addressed as the whole lattice
    float* y;                      // output, addressed as the whole lattice
    int k_begin, k_end;            // planes to produce
    int xk_lo, xk_hi;              // planes of x that exist on this device (inclusive)
    int L, nchunk, nxt;            // rows per walk, walks per plane, workgroups per row
    float add;                     // MODE 1: constant added to every output
    double* dot_partial;           // MODE 0: sum of x*y per workgroup [(k - k_begin) * nchunk * nxt + ...], or null
};

template <int R>
struct RbfWalkRows {
    int n;
    signed char dk[(2 * R + 1) * (2 * R + 1)], dj[(2 * R + 1) * (2 * R + 1)], dmax[(2 * R + 1) * (2 * R + 1)];
    signed char index[2 * R + 1][2 * R + 1];
};
template <int R, int D2>
constexpr RbfWalkRows<R> rbf_walk_rows()
{
    RbfWalkRows<R> r{};
    r.n = 0;
    for (int dk = -R; dk <= R; ++dk)
        for (int dj = -R; dj <= R; ++dj) {
            r.index[dk + R][dj + R] = -1;
            if (dk * dk + dj * dj > D2) continue;
            int m = 0;
            while (m + 1 <= R && dk * dk + dj * dj + (m + 1) * (m + 1) <= D2) ++m;
            r.index[dk + R][dj + R] = (signed char)r.n;
            r.dk[r.n] = (signed char)(dk + R); r.dj[r.n] = (signed char)(dj + R); r.dmax[r.n] = (signed char)m;
            r.n++;
        }
    return r;
}

// the coefficient tables in the layout of the walk: a = fastest
template <int MODE>
__global__ void __launch_bounds__(256) rbf_walk_table_kernel(const RbfLutVals* __restrict__ Vp, void* __restrict__ Tout, int nv)
{
    const RbfLutVals& V = *Vp;
    __shared__ double etab[64];
    if (threadIdx.x < 64) etab[threadIdx.x] = c_exp2_neg_64[threadIdx.x];
    __syncthreads();
    const int W = 2 * V.R + 1;
    const int64_t n = (int64_t)W * W * W * nv * nv * nv;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t t0 = t;
    const int a = (int)(t % nv); t /= nv;
    const int di = (int)(t % W); t /= W;
    const int b = (int)(t % nv); t /= nv;
    const int dj = (int)(t % W); t /= W;
    const int c = (int)(t % nv); t /= nv;
    const int dk = (int)t;
    const float dx = V.v[0][di][a], dy = V.v[1][dj][b], dz = V.v[2][dk][c];
    const float r = sqrtf(dx * dx + dy * dy + dz * dz);
    if (MODE == 0) {   // rbf_matvec_kernel's arithmetic (unused variant slots hold NaN: comparison false -> 0)
        const double u = (double)r / V.sigma;
        const double val = exp(-(u * u));
        ((float*)Tout)[t0] = (val > V.thr) ? (float)val : 0.0f;
    } else {           // rbf_apply_point's arithmetic
        const double inv_sigma = 1.0 / V.sigma;
        double val = 0.0;
        if (r <= V.max_distance) {
            const double u = (double)r * inv_sigma;
            val = exp_neg_fast(u * u, etab);
        }
        ((double*)Tout)[t0] = val;
    }
}

typedef unsigned int rbf_u32x4 __attribute__((ext_vector_type(4)));
typedef float rbf_f32x4 __attribute__((ext_vector_type(4)));

// lane i-1 / lane i+1 of the wavefront (0 shifted in at its ends)
__device__ __forceinline__ float rbf_dpp_prev(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));   // wave_shr:1
}
__device__ __forceinline__ float rbf_dpp_next(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));   // wave_shl:1
}

// DPP = 1 (MODE 0): a lane keeps only ITS column of the window (25 registers instead of 125, one load per new row
// instead of 2R+1) and takes x[i +- 1], x[i +- 2] from the neighbouring lanes by DPP wavefront shifts folded into the
// multiplications; the first and last R lanes of a wavefront are halo lanes (they load and shift, but own no output), so a
// wavefront produces 64 - 2R outputs.  A third of the registers = twice the wavefronts per SIMD, which is what this
// instruction-issue-bound kernel needs (one wavefront issues a VALU instruction every 4 cycles at best).
template <int R, int D2, int MODE, int NV, int NW, int DPP>
struct RbfWalk {
    static constexpr int W = 2 * R + 1;
    static constexpr int WS = DPP ? 1 : W;                // shifted copies kept per window row
    static constexpr int OUTW = DPP ? 64 - 2 * R : 64;    // outputs per wavefront
    typedef typename std::conditional<MODE == 0, float, double>::type TE;
    static constexpr int ES = (int)sizeof(TE);
    static constexpr int NT = NW * 64;
    static constexpr int ROWLEN = W * NV;
    static constexpr int NROW = rbf_walk_rows<R, D2>().n;
    static constexpr int CPR = ROWLEN * ES / 16;          // 16-byte chunks per row of T
    static constexpr int NCHUNK = NROW * CPR;
    static constexpr int NROUND = (NCHUNK + NT - 1) / NT;
    // LDS image of one step's coefficients.
    //   MODE 1: as in T, [row][di][a] doubles (one 8-byte read per neighbour, in the neighbours' order).
    //   MODE 0: transposed, [di][a][row] floats - the rows in the order of the row sum - so that ONE 16-byte read gives a
    //           lane its coefficients of four consecutive rows (26 reads per output instead of 81; the LDS pipe, 8
    //           wavefronts per CU reading their own coefficients, is what bounds this kernel).  RS floats per (di, a): 28 =
    //           4 x odd puts the 16 variants of a 16-lane read group on disjoint banks; BS floats per di: RS NV + 4 spreads
    //           the staging stores.
    static constexpr int NQ4 = (NROW + 3) / 4;            // 16-byte groups of rows
    static constexpr int RS = 4 * NQ4 + 4 + ((NQ4 + 1) % 2 ? 0 : 4);
    static constexpr int GR = DPP ? 2 : 4;                // rows per read: 16-byte reads, or 8-byte ones where registers are short
    static constexpr int NQ = (NROW + GR - 1) / GR;
    typedef float wq_t __attribute__((ext_vector_type(GR)));
    static constexpr int BS = RS * NV + 4;
    static constexpr int BUFB = MODE == 0 ? W * BS * 4 : NROW * ROWLEN * ES;

    struct State {
        float xw[W][W][WS];       // [plane][ring slot][shift] (DPP: the lane's own column only)
        uint32_t aoff[W];         // byte offset of the lane's x variant inside the LDS image, per di
        uint32_t xoff[W];         // byte offset of the lane's (clamped) column i + di inside a row of x
        int kp[W];                // plane k + dk, clamped into the planes that exist, relative to the descriptor's base
        uint32_t gpre[NROUND];    // chunk (16 bytes) offset of the lane's staging chunks in T without the row variant b
        uint32_t sbad[NROUND];    // ... where their b is found in sB
        uint32_t sdst[NROUND];    // ... and where they go in the LDS image
        rbf_u32x4 tv[NROUND];     // staged chunks in flight
        __amdgpu_buffer_rsrc_t rx;
    };

    __device__ static __forceinline__ void load_row(const RbfWalkArgs& A, State& S, int p, int slot, int jrow)
    {
        const int jj = jrow < 0 ? 0 : (jrow >= A.ny ? A.ny - 1 : jrow);
        const int soff = (S.kp[p] * A.ny + jj) * A.nx * 4;
        if constexpr (DPP) {
            S.xw[p][slot][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(S.rx, (int)S.xoff[R], soff, 0));
        } else {
#pragma unroll
            for (int d = 0; d < W; ++d)
                S.xw[p][slot][d] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(S.rx, (int)S.xoff[d], soff, 0));
        }
    }
    __device__ static __forceinline__ void table_fetch(const RbfWalkArgs& A, State& S, const uint8_t* sB, int sn)
    {
        const rbf_u32x4* __restrict__ T = (const rbf_u32x4*)A.T;
#pragma unroll
        for (int r = 0; r < NROUND; ++r) {
            const uint32_t b = sB[S.sbad[r] + (uint32_t)sn];
            S.tv[r] = T[S.gpre[r] + b * (uint32_t)CPR];
        }
    }
    __device__ static __forceinline__ void table_store(State& S, char* sT, int buf, uint32_t tid)
    {
#pragma unroll
        for (int r = 0; r < NROUND; ++r) {
            const uint32_t e = tid + (uint32_t)(r * NT);
            if ((r + 1) * NT <= NCHUNK || e < (uint32_t)NCHUNK) {
                char* dst = sT + buf * BUFB + S.sdst[r];
                if constexpr (MODE == 0) {   // four x variants of one (row, di): RS floats apart
                    *(uint32_t*)(dst) = S.tv[r].x;
                    *(uint32_t*)(dst + RS * 4) = S.tv[r].y;
                    *(uint32_t*)(dst + 2 * RS * 4) = S.tv[r].z;
                    *(uint32_t*)(dst + 3 * RS * 4) = S.tv[r].w;
                } else {
                    *(rbf_u32x4*)dst = S.tv[r];
                }
            }
        }
    }
    // where chunk e of a step's coefficients (16 bytes of T: row e / CPR, then di, then a) goes in the LDS image
    __device__ static __forceinline__ uint32_t chunk_dst(uint32_t e)
    {
        if constexpr (MODE == 0) {
            const uint32_t row = e / (uint32_t)CPR, within = e - row * (uint32_t)CPR;   // CPR = W NV / 4
            const uint32_t di = within / (uint32_t)(NV / 4), a4 = within - di * (uint32_t)(NV / 4);
            return ((di * (uint32_t)BS + 4u * a4 * (uint32_t)RS) + row) * 4u;
        } else {
            return e * 16u;
        }
    }
    __device__ static __forceinline__ uint32_t lane_aoff(uint32_t di, uint32_t a)
    {
        return MODE == 0 ? (di * (uint32_t)BS + a * (uint32_t)RS) * 4u : a * (uint32_t)ES;
    }

    struct Ctx {   // the workgroup's / lane's place in the lattice
        int k, j0, Lc, i;
        bool valid;
    };
    template <int P>
    __device__ static __forceinline__ float step(const RbfWalkArgs& A, State& S, const char* sT, int s, const Ctx& C)
    {
        const int j0 = C.j0;
        constexpr RbfWalkRows<R> RL = rbf_walk_rows<R, D2>();
        const int jnew = j0 + s + R + 1;   // the row that enters the window for the NEXT step (into the slot of dj = -R)
        uint32_t ab[W];
#pragma unroll
        for (int d = 0; d < W; ++d) ab[d] = S.aoff[d] + (uint32_t)((s & 1) * BUFB);
        float acc = 0.0f;
        if constexpr (MODE == 0) {
            // software pipeline over groups of four rows: the reads of group g + 1 are issued, then the products of
            // group g run (the compiler's own order - every read, one wait, the serial sum - leaves the LDS pipe and the
            // VALU idle in turn, in all wavefronts of the workgroup at once)
            wq_t wq[2][W];
            auto read_group = [&](int g, wq_t (&dst)[W]) {
#pragma unroll
                for (int di = 0; di < W; ++di) {
                    bool used = false;
#pragma unroll
                    for (int rr = 0; rr < GR; ++rr)
                        if (GR * g + rr < NROW && RL.dmax[GR * g + rr] >= (di > R ? di - R : R - di)) used = true;
                    if (used) dst[di] = *(const wq_t*)(sT + ab[di] + g * (GR * 4));
                }
            };
            read_group(0, wq[0]);
#pragma unroll
            for (int g = 0; g < NQ; ++g) {
                if (g + 1 < NQ) read_group(g + 1, wq[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < GR; ++rr) {
                    const int row = GR * g + rr;
                    if (row >= NROW) continue;
                    const int p = RL.dk[row], dj = RL.dj[row], m = RL.dmax[row];
                    if constexpr (DPP) {
                        static_assert(!DPP || R == 2, "the DPP form is written for R = 2");
                        const float own = S.xw[p][(P + dj) % W][0];
                        const wq_t* wv = wq[g & 1];
                        if (m == 2) {
                            const float t1 = rbf_dpp_prev(own), t2 = rbf_dpp_next(own);
                            acc += wv[0][rr] * rbf_dpp_prev(t1);
                            acc += wv[1][rr] * t1;
                            acc += wv[2][rr] * own;
                            acc += wv[3][rr] * t2;
                            acc += wv[4][rr] * rbf_dpp_next(t2);
                        } else if (m == 1) {
                            acc += wv[1][rr] * rbf_dpp_prev(own);
                            acc += wv[2][rr] * own;
                            acc += wv[3][rr] * rbf_dpp_next(own);
                        } else {
                            acc += wv[2][rr] * own;
                        }
                    } else {
#pragma unroll
                        for (int di = R - m; di <= R + m; ++di) acc += wq[g & 1][di][rr] * S.xw[p][(P + dj) % W][di];
                    }
                    // the plane's last row has been used: the slot of its dj = -R row takes the row of the next step
                    if (row + 1 == NROW || RL.dk[row + 1] != p) load_row(A, S, p, P % W, jnew);
                }
                asm volatile("" : "+v"(acc));   // (pins the products of the group between the two barriers)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            static_assert(MODE == 0 || !DPP, "the evaluation keeps the shifted copies");
            constexpr RbfTapOrder<R> TO = rbf_tap_order<R, D2>();
            constexpr int NB = 9;   // neighbours per batch
            double eb[2][NB];
            auto read_batch = [&](int q0, double (&dst)[NB]) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int q = q0 + u;
                    if (q >= TO.n) continue;
                    const int row = RL.index[TO.dk[q]][TO.dj[q]];
                    dst[u] = ((const double*)(sT + ab[TO.di[q]]))[(row * W + TO.di[q]) * NV];
                }
            };
            read_batch(0, eb[0]);
#pragma unroll
            for (int q0 = 0; q0 < TO.n; q0 += NB) {
                if (q0 + NB < TO.n) read_batch(q0 + NB, eb[(q0 / NB + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int q = q0 + u;
                    if (q >= TO.n) continue;
                    const int p = TO.dk[q], dj = TO.dj[q], di = TO.di[q];
                    // (the conversion as an opaque instruction: left to the compiler, the Float64 images of the window's
                    //  values are kept across the steps of the unrolled ring - twice the registers of the window)
                    double wd;
                    asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(wd) : "v"(S.xw[p][(P + dj) % W][DPP ? 0 : di]));
                    acc = (float)((double)acc + wd * eb[(q0 / NB) & 1][u]);
                }
                asm volatile("" : "+v"(acc));   // (pins the sums of the batch between the two barriers)
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < W; ++p) load_row(A, S, p, P % W, jnew);
        }
        return acc;
    }
};

// (DPP: at most 96 registers, so that two workgroups of nine wavefronts share a CU)
template <int R, int D2, int MODE, int NV, int NW, int DPP>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(DPP && NW >= 3 ? 5 : 1))) rbf_walk_kernel(const RbfWalkArgs A)
{
    typedef RbfWalk<R, D2, MODE, NV, NW, DPP> K;
    typedef typename K::TE TE;
    constexpr int W = K::W, NT = K::NT, NROUND = K::NROUND;
    constexpr RbfWalkRows<R> RL = rbf_walk_rows<R, D2>();
    __shared__ __attribute__((aligned(16))) char sT[2 * K::BUFB];
    __shared__ uint8_t sB[W * RBF_WALK_LMAX];
    const uint32_t tid = threadIdx.x;
    const int U = A.nchunk * A.nxt;
    const int kk = (int)blockIdx.x / U, u = (int)blockIdx.x - kk * U;
    const int chunk = u / A.nxt, xt = u - chunk * A.nxt;
    const int k = A.k_begin + kk;
    const int j0 = chunk * A.L;
    const int Lc = A.ny - j0 < A.L ? A.ny - j0 : A.L;
    // lane -> column: consecutive lanes = consecutive columns; with DPP the wavefronts overlap by their 2R halo lanes
    const int lane = (int)(tid & 63u), wv = (int)(tid >> 6);
    const int i = xt * (NW * K::OUTW) + wv * K::OUTW + lane - (DPP ? R : 0);
    const bool valid = i < A.nx && (!DPP || (lane >= R && lane < 64 - R));   // (i >= 0 for lane >= R)
    const int iv = i < 0 ? 0 : (i < A.nx ? i : A.nx - 1);
    typename K::State S;
    // ---- per-lane constants ----
#pragma unroll
    for (int d = 0; d < W; ++d) {
        const uint32_t va = A.vx[d * A.nx + iv];
        S.aoff[d] = K::lane_aoff((uint32_t)d, va != 255u ? va : (uint32_t)(NV - 1));
        int ic = iv + d - R;
        ic = ic < 0 ? 0 : (ic >= A.nx ? A.nx - 1 : ic);
        S.xoff[d] = (uint32_t)ic * 4u;
    }
    int kbase = k - R;
    kbase = kbase < A.xk_lo ? A.xk_lo : (kbase > A.xk_hi ? A.xk_hi : kbase);
#pragma unroll
    for (int p = 0; p < W; ++p) {
        int kq = k + p - R;
        kq = kq < A.xk_lo ? A.xk_lo : (kq > A.xk_hi ? A.xk_hi : kq);
        S.kp[p] = kq - kbase;
    }
    const int64_t plane = (int64_t)A.nx * A.ny;
    {
        const int64_t avail = (int64_t)(A.xk_hi - kbase + 1) * plane * 4, want = (int64_t)W * plane * 4;
        S.rx = __builtin_amdgcn_make_buffer_rsrc((void*)(A.x + (int64_t)kbase * plane), 0, (int)(avail < want ? avail : want), 0x00020000);
    }
    // ---- row variants of the walk, staging bookkeeping ----
    for (int e = (int)tid; e < W * Lc; e += NT) {
        const int d = e / Lc, s = e - d * Lc;
        const uint32_t vb = A.vy[d * A.ny + j0 + s];
        sB[d * RBF_WALK_LMAX + s] = (uint8_t)(vb != 255u ? vb : (uint32_t)(NV - 1));
    }
#pragma unroll
    for (int r = 0; r < NROUND; ++r) {
        uint32_t e = tid + (uint32_t)(r * NT);
        if (e >= (uint32_t)K::NCHUNK) e = 0;   // (never stored)
        const int row = (int)(e / (uint32_t)K::CPR), within = (int)e - row * K::CPR;
        int dk = 0, dj = 0;
#pragma unroll
        for (int q = 0; q < RL.n; ++q)
            if (row == q) { dk = RL.dk[q]; dj = RL.dj[q]; }
        const uint32_t vc = A.vz[dk * A.nz + k];
        const uint32_t c = vc != 255u ? vc : (uint32_t)(NV - 1);
        S.gpre[r] = (((uint32_t)dk * NV + c) * W + (uint32_t)dj) * (uint32_t)(NV * K::CPR) + (uint32_t)within;
        S.sbad[r] = (uint32_t)(dj * RBF_WALK_LMAX);
        S.sdst[r] = K::chunk_dst(e);
    }
    __syncthreads();
    K::table_fetch(A, S, sB, 0);
    K::table_store(S, sT, 0, tid);
    if (Lc > 1) K::table_fetch(A, S, sB, 1);
    // ---- the window of step 0 ----
#pragma unroll
    for (int rr = 0; rr < W; ++rr)
#pragma unroll
        for (int p = 0; p < W; ++p) K::load_row(A, S, p, rr, j0 + rr - R);
    float* __restrict__ yrow = A.y + ((int64_t)k * A.ny + j0) * A.nx + i;
    double dsum = 0.0;
    const typename K::Ctx ctx = {k, j0, Lc, i, valid};
    int s = 0;
#define RBF_WALK_STEP(P)                                                                                        \
    {                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                         \
        if (s + 1 < Lc) K::table_store(S, sT, (s + 1) & 1, tid);                                                \
        if (s + 2 < Lc) K::table_fetch(A, S, sB, s + 2);                                                        \
        const float xc = S.xw[R][((P) + R) % W][DPP ? 0 : R];                                                   \
        const float acc = K::template step<(P)>(A, S, sT, s, ctx);                                              \
        asm volatile("" ::"v"(acc)); /* (keeps the sum where it is: otherwise it sinks into the branch of the store) */ \
        if (valid) {                                                                                            \
            yrow[(int64_t)s * A.nx] = MODE == 0 ? acc : acc + A.add;                                            \
            if (MODE == 0) dsum += (double)xc * (double)acc;                                                    \
        }                                                                                                       \
        if (++s >= Lc) break;                                                                                   \
    }
    for (;;) {
        RBF_WALK_STEP(0)
        if constexpr (W > 1) RBF_WALK_STEP(1)
        if constexpr (W > 2) RBF_WALK_STEP(2)
        if constexpr (W > 3) RBF_WALK_STEP(3)
        if constexpr (W > 4) RBF_WALK_STEP(4)
        if constexpr (W > 5) RBF_WALK_STEP(5)
        if constexpr (W > 6) RBF_WALK_STEP(6)
    }
#undef RBF_WALK_STEP
    if (MODE == 0 && A.dot_partial) {
        // sum of x * y over the workgroup's outputs in a fixed order: lanes (xor tree), then wavefronts
        __shared__ double sred[NW];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off, 64);
        if ((tid & 63u) == 0) sred[tid >> 6] = dsum;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NW; ++w) t += sred[w];
            A.dot_partial[blockIdx.x] = t;
        }
    }
}

// dot(a, b) as the partial sums the product kernel forms for dot(x, y) (RbfWalkArgs::dot_partial): per workgroup of the walk
// (plane k, rows [j0, j0 + L), 60 NW columns) every lane adds its column's products over the rows in order (Float64), then
// the lanes of a wavefront (xor butterfly), then the wavefronts in order.  partial[(k - k_begin) nchunk nxt + chunk nxt + xt].
template <int NW>
__global__ void __launch_bounds__(NW * 64) rbf_walk_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int nx, int ny,
                                                              int k_begin, int L, int nchunk, int nxt, double* __restrict__ partial)
{
    constexpr int R = 2, OUTW = 64 - 2 * R;
    const uint32_t tid = threadIdx.x;
    const int U = nchunk * nxt;
    const int kk = (int)blockIdx.x / U, u = (int)blockIdx.x - kk * U;
    const int chunk = u / nxt, xt = u - chunk * nxt;
    const int k = k_begin + kk, j0 = chunk * L;
    const int Lc = ny - j0 < L ? ny - j0 : L;
    const int lane = (int)(tid & 63u), wv = (int)(tid >> 6);
    const int i = xt * (NW * OUTW) + wv * OUTW + lane - R;
    const bool valid = i < nx && lane >= R && lane < 64 - R;
    double dsum = 0.0;
    if (valid) {
        const int64_t o = ((int64_t)k * ny + j0) * nx + i;
        for (int s = 0; s < Lc; ++s) dsum += (double)a[o + (int64_t)s * nx] * (double)b[o + (int64_t)s * nx];
    }
    __shared__ double sred[NW];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off, 64);
    if ((tid & 63u) == 0) sred[tid >> 6] = dsum;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += sred[w];
        partial[blockIdx.x] = t;
    }
}

// ---- host side ----
struct RbfWalkPlan {
    int NW, L, nchunk, nxt;
};
// dpp: wavefronts of 60 outputs, NW in {1, 2, 3, 5, 9}; else 64 outputs, NW in {1, 2, 4, 8}
// (nplanes: the planes of the WHOLE lattice - a Z-slab of a multi-device run cuts its walks like the single-device run, so
//  that the partial sums of the dot product come out the same)
static RbfWalkPlan rbf_walk_plan(int nx, int ny, int nplanes, bool dpp, int stage_chunks = 420)
{
    static const int nw_dpp[] = {1, 2, 3, 5, 9}, nw_std[] = {1, 2, 4, 8};
    const int* cand = dpp ? nw_dpp : nw_std;
    const int ncand = dpp ? 5 : 4, outw = dpp ? 60 : 64;
    RbfWalkPlan P;
    // wavefronts per workgroup: the cheapest row.  A wavefront-step costs 1; the table rows of a step are staged once per
    // workgroup (16-byte chunks: 420 for the product, 840 / 3 360 for the evaluation), ~0.12 per round of a thread -
    // narrow workgroups stage the same rows again for every piece of a row (nx = 257: one wavefront per workgroup took
    // 0.68 ms where eight take less although half their lanes idle)
    P.NW = cand[0];
    double best = -1.0;
    for (int c = 0; c < ncand; ++c) {
        const int nw = cand[c];
        const double waves = (double)((nx + outw * nw - 1) / (outw * nw)) * nw;
        const double cost = waves * (1.0 + 0.12 * (double)((stage_chunks + 64 * nw - 1) / (64 * nw)));
        if (best < 0.0 || cost <= best) { best = cost; P.NW = nw; }
    }
    if (const char* e = getenv("R2S_RBF_WALK_NW")) {   // (experiments)
        const int nw = atoi(e);
        for (int c = 0; c < ncand; ++c)
            if (cand[c] == nw) P.NW = nw;
    }
    P.nxt = (nx + outw * P.NW - 1) / (outw * P.NW);
    // rows per walk: long walks amortise the window of the first step, short ones fill the chip on small lattices
    P.L = RBF_WALK_LMAX;
    while (P.L > 4 && (int64_t)nplanes * ((ny + P.L - 1) / P.L) * P.nxt < 2048) P.L /= 2;
    P.nchunk = (ny + P.L - 1) / P.L;
    return P;
}
// is there a walk kernel for this stencil / lattice?
static bool rbf_walk_supported(int R, int tap_d2, int nx, int ny)
{
    return R == 2 && tap_d2 == 7 && (int64_t)nx * ny * 5 * 4 < 0x7FFFFFFFll;
}
template <int MODE, int NV>
static void rbf_walk_launch_nw(const RbfWalkArgs& A, int NW, unsigned nb, hipStream_t st)
{
    switch (NW) {
    case 1: rbf_walk_kernel<2, 7, MODE, NV, 1, 0><<<nb, 64, 0, st>>>(A); break;
    case 2: rbf_walk_kernel<2, 7, MODE, NV, 2, 0><<<nb, 128, 0, st>>>(A); break;
    case 4: rbf_walk_kernel<2, 7, MODE, NV, 4, 0><<<nb, 256, 0, st>>>(A); break;
    default: rbf_walk_kernel<2, 7, MODE, NV, 8, 0><<<nb, 512, 0, st>>>(A); break;
    }
}
static void rbf_walk_launch_dpp(const RbfWalkArgs& A, int NW, unsigned nb, hipStream_t st)
{
    switch (NW) {
    case 1: rbf_walk_kernel<2, 7, 0, 16, 1, 1><<<nb, 64, 0, st>>>(A); break;
    case 2: rbf_walk_kernel<2, 7, 0, 16, 2, 1><<<nb, 128, 0, st>>>(A); break;
    case 3: rbf_walk_kernel<2, 7, 0, 16, 3, 1><<<nb, 192, 0, st>>>(A); break;
    case 5: rbf_walk_kernel<2, 7, 0, 16, 5, 1><<<nb, 320, 0, st>>>(A); break;
    default: rbf_walk_kernel<2, 7, 0, 16, 9, 1><<<nb, 576, 0, st>>>(A); break;
    }
}
// mode 0: y = K x (T: float table, nv = 16); mode 1: evaluation (T: double table, nv = 16 or 64)
static void rbf_walk_launch(int mode, int nv, RbfWalkArgs A, hipStream_t st)
{
    const bool dpp = mode == 0;
    const RbfWalkPlan P = rbf_walk_plan(A.nx, A.ny, A.nz, dpp, mode == 0 ? 420 : (nv == 16 ? 840 : 3360));
    A.L = P.L; A.nchunk = P.nchunk; A.nxt = P.nxt;
    const unsigned nb = (unsigned)((int64_t)(A.k_end - A.k_begin) * P.nchunk * P.nxt);
    if (nb == 0) return;
    if (dpp) rbf_walk_launch_dpp(A, P.NW, nb, st);
    else if (nv == 16) rbf_walk_launch_nw<1, 16>(A, P.NW, nb, st);
    else rbf_walk_launch_nw<1, 64>(A, P.NW, nb, st);
}

// workgroups (= partial sums of the dot product) of the planes [k_begin, k_end)
static size_t rbf_walk_nparts(int nx, int ny, int nz, int k_begin, int k_end)
{
    const RbfWalkPlan P = rbf_walk_plan(nx, ny, nz, true);
    return (size_t)(k_end - k_begin) * P.nchunk * P.nxt;
}
static void rbf_walk_dot_launch(const float* a, const float* b, int nx, int ny, int nz, int k_begin, int k_end, double* partial, hipStream_t st)
{
    const RbfWalkPlan P = rbf_walk_plan(nx, ny, nz, true);
    const unsigned nb = (unsigned)((size_t)(k_end - k_begin) * P.nchunk * P.nxt);
    if (nb == 0) return;
    switch (P.NW) {
    case 1: rbf_walk_dot_kernel<1><<<nb, 64, 0, st>>>(a, b, nx, ny, k_begin, P.L, P.nchunk, P.nxt, partial); break;
    case 2: rbf_walk_dot_kernel<2><<<nb, 128, 0, st>>>(a, b, nx, ny, k_begin, P.L, P.nchunk, P.nxt, partial); break;
    case 3: rbf_walk_dot_kernel<3><<<nb, 192, 0, st>>>(a, b, nx, ny, k_begin, P.L, P.nchunk, P.nxt, partial); break;
    case 5: rbf_walk_dot_kernel<5><<<nb, 320, 0, st>>>(a, b, nx, ny, k_begin, P.L, P.nchunk, P.nxt, partial); break;
    default: rbf_walk_dot_kernel<9><<<nb, 576, 0, st>>>(a, b, nx, ny, k_begin, P.L, P.nchunk, P.nxt, partial); break;
    }
}
