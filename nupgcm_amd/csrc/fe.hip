// Element-local finite-element kernels: the per-timestep advection right-hand side (src/model.jl:269-278), the
// (re)assembly of M, Kh, Kv, A, B into fixed CSR patterns (src/evolution.jl:209-264, src/inversion.jl:133-219), the
// diffusion right-hand side (src/evolution.jl:269-278), the coefficient closures of src/inputs.jl:87-91,130-137 and the
// CFL reduction of src/timesteppers.jl:108-119.  In the reference all of these are single-threaded Gridap cell loops on
// the host, also in GPU mode.
//
// Data layout (HBM): every per-cell table is stored transposed, [component][cell], so that thread-per-cell kernels read
// it with unit stride across the 64 lanes of a wave.  The quadrature weights and the P2/P1 shape-function tables
// (at most 16 x (10 + 40 + 4) doubles) are staged once per workgroup in LDS; all lanes of a wave read the same table
// entry in the same instruction, which the LDS serves as a broadcast.
//
// The vector assembly is deterministic: pass 1 writes each cell's local vector to loc[i][cell]; pass 2 gives every
// destination DoF to one thread which adds its contributions in a fixed (cell-ascending) order through an inverted index
// - no atomics, bit-reproducible right-hand sides.  The matrix (re)assembly (set-up and coefficient refreshes) is
// deterministic too: one thread per CSR row walks the (cell, local DoF) pairs that carry the row's DoF, cell-ascending,
// recomputes the local rows it owns and adds them into its own row (slots found by binary search in the sorted row).
#include <algorithm>
#include <cmath>
#include <numeric>

#include "common.h"
#include "device_utils.h"

namespace npg {

constexpr int kMaxQ = 16;

struct FeDev {
    int64_t ncell;
    int nq, nb;                 // nb = buoyancy nodes per cell (10 or 4)
    const double *G;            // [12][ncell]   grad lambda_k, component a at (3k+a)
    const double *wdet;         // [ncell]
    const double *qw, *N2, *dN2, *Nb, *dNb, *N1;
    const int32_t *cu;          // [30][ncell]  (3*i + a)
    const int32_t *cp;          // [4][ncell]
    const int32_t *cb;          // [nb][ncell]
    const double *u_diri, *b_diri;
    const double *nu, *kh, *kv, *f;   // [nq][ncell] or null
};

// R = the arithmetic type of the element-LOCAL work (shape tables, geometry, nodal values, the integrand at a quadrature
// point): double, or float for the mixed mode of BASELINE.json configs[4] ("fp32 assembly / fp64 solve").  Whatever R is,
// sums over quadrature points, over the cells of a row and everything downstream (CSR values, right-hand sides, solvers)
// are fp64, and all HBM tables stay fp64 (converted on load).
template <typename R>
struct FeTablesT {
    R qw[kMaxQ];
    R N2[kMaxQ * 10];
    R dN2[kMaxQ * 40];
    R Nb[kMaxQ * 10];
    R dNb[kMaxQ * 40];
    R N1[kMaxQ * 4];
};
using FeTables = FeTablesT<double>;

template <typename R>
__device__ __forceinline__ void stage_tables(const FeDev &d, FeTablesT<R> &t) {
    for (int i = threadIdx.x; i < d.nq; i += blockDim.x) t.qw[i] = (R)d.qw[i];
    for (int i = threadIdx.x; i < d.nq * 10; i += blockDim.x) t.N2[i] = (R)d.N2[i];
    for (int i = threadIdx.x; i < d.nq * 40; i += blockDim.x) t.dN2[i] = (R)d.dN2[i];
    for (int i = threadIdx.x; i < d.nq * d.nb; i += blockDim.x) t.Nb[i] = (R)d.Nb[i];
    for (int i = threadIdx.x; i < d.nq * d.nb * 4; i += blockDim.x) t.dNb[i] = (R)d.dNb[i];
    for (int i = threadIdx.x; i < d.nq * 4; i += blockDim.x) t.N1[i] = (R)d.N1[i];
    __syncthreads();
}

__device__ __forceinline__ double field_val(const double *x, const double *diri, int32_t idx) {
    return idx >= 0 ? x[idx] : diri[-1 - idx];
}

// ---- advection: pass 1 ------------------------------------------------------------------------------------------------
// loc[i][cell] = int ( c1 b + c2 b_prev - cdt ( u~ . grad b~ + u~_z N2 ) ) phi_i     (src/model.jl:292-300)
template <typename R, int NB>
__global__ void __launch_bounds__(kBlock) k_advection_local(FeDev d, int scheme, double dt, double N2, const double *b,
                                                            const double *bp, const double *xi, const double *xip,
                                                            double *loc) {
    __shared__ FeTablesT<R> t;
    stage_tables(d, t);
    const int64_t cell = blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (cell >= d.ncell) return;
    const bool bdf2 = scheme == NPG_BDF2;
    const double c1 = bdf2 ? 4.0 / 3.0 : 1.0, c2 = bdf2 ? -1.0 / 3.0 : 0.0, e1 = bdf2 ? 2.0 : 1.0,
                 e2 = bdf2 ? -1.0 : 0.0;
    const R cdt = (R)(bdf2 ? 2.0 / 3.0 * dt : dt), rN2 = (R)N2;
    R G[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) G[k] = (R)d.G[(size_t)k * d.ncell + cell];
    const R wdet = (R)d.wdet[cell];
    R bm[NB], bt[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int32_t idx = d.cb[(size_t)i * d.ncell + cell];
        const double v = field_val(b, d.b_diri, idx), vp = field_val(bp, d.b_diri, idx);
        bm[i] = (R)(c1 * v + c2 * vp);         // the BDF combinations of the fp64 state are formed before rounding
        bt[i] = (R)(e1 * v + e2 * vp);
    }
    R ut[30];
#pragma unroll
    for (int k = 0; k < 30; ++k) {
        const int32_t idx = d.cu[(size_t)k * d.ncell + cell];
        ut[k] = (R)(e1 * field_val(xi, d.u_diri, idx) + e2 * field_val(xip, d.u_diri, idx));
    }
    double acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = 0.0;
    for (int q = 0; q < d.nq; ++q) {
        R bq = 0, gl0 = 0, gl1 = 0, gl2 = 0, gl3 = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            bq += t.Nb[q * NB + i] * bm[i];
            const R *dn = &t.dNb[(q * NB + i) * 4];
            gl0 += dn[0] * bt[i];
            gl1 += dn[1] * bt[i];
            gl2 += dn[2] * bt[i];
            gl3 += dn[3] * bt[i];
        }
        R ux = 0, uy = 0, uz = 0;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const R n = t.N2[q * 10 + i];
            ux += n * ut[3 * i];
            uy += n * ut[3 * i + 1];
            uz += n * ut[3 * i + 2];
        }
        const R gx = gl0 * G[0] + gl1 * G[3] + gl2 * G[6] + gl3 * G[9];
        const R gy = gl0 * G[1] + gl1 * G[4] + gl2 * G[7] + gl3 * G[10];
        const R gz = gl0 * G[2] + gl1 * G[5] + gl2 * G[8] + gl3 * G[11];
        const R integrand = bq - cdt * (ux * gx + uy * gy + uz * gz + uz * rN2);
        const R wq = t.qw[q] * wdet * integrand;
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[i] += (double)(wq * t.Nb[q * NB + i]);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) loc[(size_t)i * d.ncell + cell] = acc[i];
}

// ---- rhs_diff: pass 1 : loc[i][cell] = -N2 int kappa_v d_z phi_i -------------------------------------------------------
template <typename R, int NB>
__global__ void __launch_bounds__(kBlock) k_rhs_diff_local(FeDev d, double N2, double *loc) {
    __shared__ FeTablesT<R> t;
    stage_tables(d, t);
    const int64_t cell = blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (cell >= d.ncell) return;
    R Gz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) Gz[k] = (R)d.G[(size_t)(3 * k + 2) * d.ncell + cell];
    const R wdet = (R)d.wdet[cell];
    double acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = 0.0;
    for (int q = 0; q < d.nq; ++q) {
        const R wq = -(R)N2 * t.qw[q] * wdet * (R)d.kv[(size_t)q * d.ncell + cell];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const R *dn = &t.dNb[(q * NB + i) * 4];
            acc[i] += (double)(wq * (dn[0] * Gz[0] + dn[1] * Gz[1] + dn[2] * Gz[2] + dn[3] * Gz[3]));
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) loc[(size_t)i * d.ncell + cell] = acc[i];
}

// ---- pass 2: fixed-order gather into the destination vector, fused with the right-hand-side combination ------------
struct RhsTerms {
    double theta, dt;
    const double *rhs_diff, *rhs_flux, *rhs_M, *rhs_h, *rhs_v;
};

__global__ void __launch_bounds__(kBlock) k_gather_rows(const int64_t *gptr, const int32_t *gidx, const double *loc,
                                                        int64_t n, RhsTerms rt, double *out) {
    for (int64_t r = blockIdx.x * (int64_t)kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
        double s = 0.0;
        for (int64_t k = gptr[r]; k < gptr[r + 1]; ++k) s += loc[gidx[k]];
        if (rt.rhs_diff) s += rt.theta * rt.rhs_diff[r];
        if (rt.rhs_flux) s += rt.dt * rt.rhs_flux[r];
        double lift = 0.0;
        if (rt.rhs_h) lift += rt.rhs_h[r];
        if (rt.rhs_v) lift += rt.rhs_v[r];
        lift *= rt.theta;
        if (rt.rhs_M) lift += rt.rhs_M[r];
        out[r] = s - lift;
    }
}

// ---- matrix assembly ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t csr_slot(const int64_t *rowptr, const int32_t *col, int32_t row, int32_t c) {
    int64_t lo = rowptr[row], hi = rowptr[row + 1] - 1;
    while (lo <= hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int32_t v = col[mid];
        if (v == c) return mid;
        if (v < c) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// a row owner adds into its own row: no atomic, contributions arrive in the fixed order of its adjacency list
__device__ __forceinline__ void row_add(const int64_t *rowptr, const int32_t *col, double *val, int32_t row, int32_t c,
                                        double v, int *missing) {
    const int64_t s = csr_slot(rowptr, col, row, c);
    if (s >= 0) val[s] += v;
    else if (v != 0.0) atomicAdd(missing, 1);    // a numerically non-zero entry outside the pattern is an error
}

// physical gradient of buoyancy basis function i at quadrature point q
template <typename R, int NB>
__device__ __forceinline__ void grad_b(const FeTablesT<R> &t, const R *G, int q, int i, R &gx, R &gy, R &gz) {
    const R *dn = &t.dNb[(q * NB + i) * 4];
    gx = dn[0] * G[0] + dn[1] * G[3] + dn[2] * G[6] + dn[3] * G[9];
    gy = dn[0] * G[1] + dn[1] * G[4] + dn[2] * G[7] + dn[3] * G[10];
    gz = dn[0] * G[2] + dn[1] * G[5] + dn[2] * G[8] + dn[3] * G[11];
}

template <typename R>
__device__ __forceinline__ void grad_u(const FeTablesT<R> &t, const R *G, int q, int i, R &gx, R &gy, R &gz) {
    const R *dn = &t.dN2[(q * 10 + i) * 4];
    gx = dn[0] * G[0] + dn[1] * G[3] + dn[2] * G[6] + dn[3] * G[9];
    gy = dn[0] * G[1] + dn[1] * G[4] + dn[2] * G[7] + dn[3] * G[10];
    gz = dn[0] * G[2] + dn[1] * G[5] + dn[2] * G[8] + dn[3] * G[11];
}

// ---- matrix assembly: sixteen lanes per CSR row, lane = quadrature point ------------------------------------------------
// A lane group owns a row.  It walks the (cell, local DoF) pairs that carry the row's DoF (an inverted index, cell-ascending);
// for each pair every lane evaluates its quadrature point's share of the local row, the shares are summed over the lanes in
// the fixed order of the DPP tree, and the sums go into the group's own CSR row (slots by binary search) - no atomics, the
// same bits on every run.  The three component rows of a velocity node repeat the friction integral: a set-up cost.
constexpr int kQL = 16;      // lanes per row = kMaxQ

// M / Kh / Kv: rows = buoyancy DoFs; gptr / gidx = the inverted index of the vector assembly (i * ncell + cell)
template <typename R, int NB>
__global__ void __launch_bounds__(kBlock) k_assemble_b(FeDev d, int which, const int64_t *gptr, const int32_t *gidx,
                                                       int64_t nrows, const int64_t *rowptr, const int32_t *col,
                                                       double *val, double *lift, int *missing) {
    __shared__ FeTablesT<R> t;
    stage_tables(d, t);
    const int64_t r = (blockIdx.x * (int64_t)kBlock + threadIdx.x) / kQL;
    const int q = threadIdx.x % kQL;
    if (r >= nrows) return;                                  // whole lane groups leave together
    const int32_t row = (int32_t)r;
    const bool on = q < d.nq;
    double lf = 0.0;
    for (int64_t k = gptr[r]; k < gptr[r + 1]; ++k) {
        const int64_t cell = gidx[k] % d.ncell;
        const int i = (int)(gidx[k] / d.ncell);
        double acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = 0.0;
        if (on) {
            R wq = t.qw[q] * (R)d.wdet[cell];
            if (which == NPG_MAT_M) {
                wq *= t.Nb[q * NB + i];
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = (double)(wq * t.Nb[q * NB + j]);
            } else {
                R G[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) G[e] = (R)d.G[(size_t)e * d.ncell + cell];
                R gix, giy, giz;
                grad_b<R, NB>(t, G, q, i, gix, giy, giz);
                wq *= (R)(which == NPG_MAT_KH ? d.kh : d.kv)[(size_t)q * d.ncell + cell];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    R gx, gy, gz;
                    grad_b<R, NB>(t, G, q, j, gx, gy, gz);
                    acc[j] = (double)(wq * (which == NPG_MAT_KH ? (gix * gx + giy * gy) : giz * gz));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = group_sum_dpp<kQL>(acc[j]);
        // lane j adds column j (distinct columns of one cell; the next cell's adds come after these in program order)
        double mine = 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) mine = (q == j) ? acc[j] : mine;
        if (q < NB) {
            const int32_t c = d.cb[(size_t)q * d.ncell + cell];
            if (c >= 0) row_add(rowptr, col, val, row, c, mine, missing);
        }
        if (q == 0) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int32_t c = d.cb[(size_t)j * d.ncell + cell];
                if (c < 0) lf += acc[j] * d.b_diri[-1 - c];
            }
        }
        __builtin_amdgcn_s_waitcnt(0);                       // this cell's stores are out before the next cell's loads
    }
    if (lift && q == 0) lift[row] = lf;
}

// B and A: rows of the inversion system.  iptr / iidx list, cell-ascending, the (cell, local DoF l) pairs that carry the row's
// DoF: l = 3 i + a for component a of velocity node i, l = 30 + m for pressure vertex m (iidx = l * ncell + cell).
//
// B: rows (u node i, component z), columns buoyancy nodes: scale * int phi_i phib_j     (src/inversion.jl:208)
template <typename R, int NB>
__global__ void __launch_bounds__(kBlock) k_assemble_B(FeDev d, double scale, const int64_t *iptr, const int32_t *iidx,
                                                       int64_t nrows, const int64_t *rowptr, const int32_t *col,
                                                       double *val, double *lift, int *missing) {
    __shared__ FeTablesT<R> t;
    stage_tables(d, t);
    const int64_t r = (blockIdx.x * (int64_t)kBlock + threadIdx.x) / kQL;
    const int q = threadIdx.x % kQL;
    if (r >= nrows) return;
    const int32_t row = (int32_t)r;
    const bool on = q < d.nq;
    double lf = 0.0;
    for (int64_t k = iptr[r]; k < iptr[r + 1]; ++k) {
        const int64_t cell = iidx[k] % d.ncell;
        const int l = (int)(iidx[k] / d.ncell);
        if (l >= 30 || l % 3 != 2) continue;                 // only the vertical momentum rows feel buoyancy (group-uniform)
        const int i = l / 3;
        double acc[NB];
        const R wq = on ? t.qw[q] * (R)d.wdet[cell] * (R)scale * t.N2[q * 10 + i] : (R)0;
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = group_sum_dpp<kQL>(on ? (double)(wq * t.Nb[q * NB + j]) : 0.0);
        double mine = 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) mine = (q == j) ? acc[j] : mine;
        if (q < NB) {
            const int32_t c = d.cb[(size_t)q * d.ncell + cell];
            if (c >= 0) row_add(rowptr, col, val, row, c, mine, missing);
        }
        if (q == 0) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int32_t c = d.cb[(size_t)j * d.ncell + cell];
                if (c < 0) lf += acc[j] * d.b_diri[-1 - c];
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (lift && q == 0) lift[row] = lf;
}

// A:
//   [(i,a),(j,c)] += a2e2 int nu ( d_ac grad phi_i . grad phi_j  [+ d_c phi_i d_a phi_j  if full_stress] )
//   [(i,x),(j,y)] -= int f phi_i phi_j ; [(i,y),(j,x)] += int f phi_i phi_j
//   [(i,a), p_m ] -= int d_a phi_i psi_m ; [p_m, (i,a)] += int psi_m d_a phi_i
template <typename R>
__global__ void __launch_bounds__(kBlock) k_assemble_A(FeDev d, double a2e2, int full_stress, const int64_t *iptr,
                                                       const int32_t *iidx, int64_t nrows, const int64_t *rowptr,
                                                       const int32_t *col, double *val, int *missing) {
    __shared__ FeTablesT<R> t;
    stage_tables(d, t);
    const int64_t r = (blockIdx.x * (int64_t)kBlock + threadIdx.x) / kQL;
    const int q = threadIdx.x % kQL;
    if (r >= nrows) return;
    const int32_t row = (int32_t)r;
    const bool on = q < d.nq;
    for (int64_t k = iptr[r]; k < iptr[r + 1]; ++k) {
        const int64_t cell = iidx[k] % d.ncell;
        const int l = (int)(iidx[k] / d.ncell);
        R G[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) G[e] = (R)d.G[(size_t)e * d.ncell + cell];
        const R wq = on ? t.qw[q] * (R)d.wdet[cell] : (R)0;
        if (l < 30) {
            const int i = l / 3, a = l % 3;
            R gi[3] = {0, 0, 0};
            R wn = 0, wf = 0;
            if (on) {
                grad_u(t, G, q, i, gi[0], gi[1], gi[2]);
                wn = wq * (R)a2e2 * (R)d.nu[(size_t)q * d.ncell + cell];
                wf = wq * (R)d.f[(size_t)q * d.ncell + cell] * t.N2[q * 10 + i];
            }
            // u-u block, one trial node j at a time; lane c < 3 then adds the entry of trial component c
            for (int j = 0; j < 10; ++j) {
                R gj[3] = {0, 0, 0};
                if (on) grad_u(t, G, q, j, gj[0], gj[1], gj[2]);
                const double kk = group_sum_dpp<kQL>((double)(wn * (gi[0] * gj[0] + gi[1] * gj[1] + gi[2] * gj[2])));
                const double cc = group_sum_dpp<kQL>(on ? (double)(wf * t.N2[q * 10 + j]) : 0.0);
                double fs[3] = {0.0, 0.0, 0.0};
                if (full_stress) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) fs[c] = group_sum_dpp<kQL>((double)(wn * gi[c] * gj[a]));
                }
                if (q < 3) {
                    const int c = q;
                    const int32_t cj = d.cu[(size_t)(3 * j + c) * d.ncell + cell];
                    if (cj >= 0) {          // homogeneous velocity Dirichlet data: no lift (src/spaces.jl u_diri_vals = 0)
                        double v = full_stress ? (c == 0 ? fs[0] : c == 1 ? fs[1] : fs[2]) : 0.0;
                        if (a == c) v += kk;
                        if (a == 0 && c == 1) v -= cc;
                        if (a == 1 && c == 0) v += cc;
                        if (a == c || full_stress || (a < 2 && c < 2)) row_add(rowptr, col, val, row, cj, v, missing);
                    }
                }
            }
            // u-p coupling of this momentum row: lane m < 4 adds the entry of pressure vertex m
            double ddm = 0.0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const double dd = group_sum_dpp<kQL>(on ? (double)(wq * t.N1[q * 4 + m] * gi[a]) : 0.0);
                ddm = (q == m) ? dd : ddm;
            }
            if (q < 4) {
                const int32_t pm = d.cp[(size_t)q * d.ncell + cell];
                if (pm >= 0) row_add(rowptr, col, val, row, pm, -ddm, missing);
            }
        } else {
            // continuity row of pressure vertex m: lane a < 3 adds the entry of component a of node i
            const int m = l - 30;
            const R wm = on ? wq * t.N1[q * 4 + m] : (R)0;
            for (int i = 0; i < 10; ++i) {
                R gi[3] = {0, 0, 0};
                if (on) grad_u(t, G, q, i, gi[0], gi[1], gi[2]);
                const double d0 = group_sum_dpp<kQL>((double)(wm * gi[0]));
                const double d1 = group_sum_dpp<kQL>((double)(wm * gi[1]));
                const double d2 = group_sum_dpp<kQL>((double)(wm * gi[2]));
                if (q < 3) {
                    const int32_t ci = d.cu[(size_t)(3 * i + q) * d.ncell + cell];
                    if (ci >= 0) row_add(rowptr, col, val, row, ci, q == 0 ? d0 : q == 1 ? d1 : d2, missing);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);                       // this cell's stores are out before the next cell's loads
    }
}

// ---- coefficient closures evaluated at the quadrature points ---------------------------------------------------------
// bz(q) = d_z b at every quadrature point, then a pointwise formula
template <int NB>
__global__ void __launch_bounds__(kBlock) k_coeff_from_bz(FeDev d, int mode, const double *b, const double *base,
                                                          double p0, double p1, double alpha, double N2, double p2,
                                                          double p3, double *out) {
    __shared__ FeTables t;
    stage_tables(d, t);
    const int64_t cell = blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (cell >= d.ncell) return;
    double Gz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) Gz[k] = d.G[(size_t)(3 * k + 2) * d.ncell + cell];
    double bn[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) bn[i] = field_val(b, d.b_diri, d.cb[(size_t)i * d.ncell + cell]);
    for (int q = 0; q < d.nq; ++q) {
        double bz = 0.0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const double *dn = &t.dNb[(q * NB + i) * 4];
            bz += bn[i] * (dn[0] * Gz[0] + dn[1] * Gz[1] + dn[2] * Gz[2] + dn[3] * Gz[3]);
        }
        const double abz = alpha * (N2 + bz);
        const size_t o = (size_t)q * d.ncell + cell;
        if (mode == 0) {
            // kappa_v + kappa_c (1 + tanh(-abz / N2min)) / 2        (src/inputs.jl:87-91)   p0 = kappa_c, p1 = N2min
            out[o] = base[o] + p0 * (1.0 + tanh(-abz / p1)) / 2.0;
        } else {
            // nu = f (f / sqrt(N2min^2 + abz^2)), then LogSumExp with nu_min    (src/inputs.jl:130-137)
            // p1 = N2min, p2 = smoothing, p3 = nu_min
            const double f = d.f[o];
            const double nu = f * (f / sqrt(p1 * p1 + abz * abz));
            const double m = fmax(p2 * p3, p2 * nu);    // stable log(exp(a) + exp(b))
            out[o] = (m + log(exp(p2 * p3 - m) + exp(p2 * nu - m))) / p2;
        }
    }
}

// ---- CFL ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_cfl(FeDev d, const double *hcell, double umin, const double *xi,
                                                double *part) {
    __shared__ FeTables t;
    __shared__ double shm[4];
    stage_tables(d, t);
    double best = 1e300;
    for (int64_t cell = blockIdx.x * (int64_t)kBlock + threadIdx.x; cell < d.ncell;
         cell += (int64_t)gridDim.x * kBlock) {
        double un[30];
#pragma unroll
        for (int k = 0; k < 30; ++k) un[k] = field_val(xi, d.u_diri, d.cu[(size_t)k * d.ncell + cell]);
        double smax = 0.0;
        for (int q = 0; q < d.nq; ++q) {
            double ux = 0.0, uy = 0.0, uz = 0.0;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const double n = t.N2[q * 10 + i];
                ux += n * un[3 * i];
                uy += n * un[3 * i + 1];
                uz += n * un[3 * i + 2];
            }
            smax = fmax(smax, sqrt(ux * ux + uy * uy + uz * uz));
        }
        best = fmin(best, hcell[cell] / fmax(smax, umin));
    }
    best = -wave_max(-best);
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmin(fmin(shm[0], shm[1]), fmin(shm[2], shm[3]));
}

}  // namespace npg

using namespace npg;

struct npg_fe {
    npg_ctx *ctx = nullptr;
    FeDev d{};
    int64_t n_inv = 0, n_b = 0;
    std::vector<void *> allocs;
    double *loc = nullptr;          // [nb][ncell]
    int64_t *gptr = nullptr;        // inverted index of the buoyancy rows (vector pass 2, matrix rows)
    int32_t *gidx = nullptr;
    int64_t *iptr = nullptr;        // inverted index of the inversion rows [u; p]: (local DoF l) * ncell + cell
    int32_t *iidx = nullptr;
    double *coef[4] = {nullptr, nullptr, nullptr, nullptr};   // nu, kappa_h, kappa_v, f
    double *kv0 = nullptr;          // background kappa_v for the convection closure
    double *hcell = nullptr;
    int *missing = nullptr;
    double *scratch_vec = nullptr;  // n_b doubles
    int precision = NPG_FE_FP64;    // arithmetic of the element-local work (npg_fe_set_precision)
};

template <typename T>
static int dev_copy(npg_fe *fe, const T *host, size_t count, const T **out) {
    T *p = nullptr;
    NPG_HIP(hipMalloc((void **)&p, std::max<size_t>(1, count) * sizeof(T)));
    fe->allocs.push_back(p);
    if (count) NPG_HIP(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = p;
    return NPG_OK;
}

// host [ncell][k] -> device [k][ncell]
template <typename T>
static int dev_copy_transposed(npg_fe *fe, const T *host, int64_t ncell, int k, const T **out) {
    std::vector<T> tmp((size_t)ncell * k);
    for (int64_t c = 0; c < ncell; ++c)
        for (int j = 0; j < k; ++j) tmp[(size_t)j * ncell + c] = host[(size_t)c * k + j];
    return dev_copy(fe, tmp.data(), tmp.size(), out);
}

NPG_API int npg_fe_create(npg_ctx *ctx, const npg_fe_desc *desc, npg_fe **out) {
    NPG_REQUIRE(ctx && desc && out, "npg_fe_create: NULL argument");
    NPG_REQUIRE(desc->ncell > 0 && desc->nq > 0 && desc->nq <= kMaxQ, "npg_fe_create: need 1 <= nq <= %d", kMaxQ);
    NPG_REQUIRE(desc->nloc_b == 10 || desc->nloc_b == 4, "npg_fe_create: nloc_b must be 10 (P2) or 4 (P1)");
    NPG_REQUIRE(desc->grad_lambda && desc->wdet && desc->qw && desc->N2 && desc->dN2 && desc->Nb && desc->dNb &&
                    desc->N1 && desc->cell_u && desc->cell_p && desc->cell_b,
                "npg_fe_create: NULL table");
    NPG_REQUIRE(desc->n_inv > 0 && desc->n_b > 0 && desc->n_inv < INT32_MAX && desc->n_b < INT32_MAX &&
                    desc->ncell * 10 < INT32_MAX,
                "npg_fe_create: sizes exceed int32 indexing");
    const int64_t nc = desc->ncell;
    const int nb = desc->nloc_b;
    // validate DoF tables on the host: every kernel indexes device vectors with them
    for (int64_t k = 0; k < nc * 30; ++k) {
        const int32_t v = desc->cell_u[k];
        NPG_REQUIRE(v < desc->n_inv && (v >= 0 || -1 - (int64_t)v < desc->n_u_diri),
                    "npg_fe_create: cell_u[%lld] = %d out of range", (long long)k, v);
    }
    for (int64_t k = 0; k < nc * 4; ++k) {
        const int32_t v = desc->cell_p[k];
        NPG_REQUIRE(v < desc->n_inv, "npg_fe_create: cell_p[%lld] = %d out of range", (long long)k, v);
    }
    for (int64_t k = 0; k < nc * nb; ++k) {
        const int32_t v = desc->cell_b[k];
        NPG_REQUIRE(v < desc->n_b && (v >= 0 || -1 - (int64_t)v < desc->n_b_diri),
                    "npg_fe_create: cell_b[%lld] = %d out of range", (long long)k, v);
    }
    npg_fe *fe = new npg_fe();
    fe->ctx = ctx;
    fe->n_inv = desc->n_inv;
    fe->n_b = desc->n_b;
    NPG_HIP(hipSetDevice(ctx->device));
    FeDev &d = fe->d;
    d.ncell = nc;
    d.nq = desc->nq;
    d.nb = nb;
    int rc;
    if ((rc = dev_copy_transposed(fe, desc->grad_lambda, nc, 12, &d.G))) return rc;
    if ((rc = dev_copy(fe, desc->wdet, (size_t)nc, &d.wdet))) return rc;
    if ((rc = dev_copy(fe, desc->qw, (size_t)desc->nq, &d.qw))) return rc;
    if ((rc = dev_copy(fe, desc->N2, (size_t)desc->nq * 10, &d.N2))) return rc;
    if ((rc = dev_copy(fe, desc->dN2, (size_t)desc->nq * 40, &d.dN2))) return rc;
    if ((rc = dev_copy(fe, desc->Nb, (size_t)desc->nq * nb, &d.Nb))) return rc;
    if ((rc = dev_copy(fe, desc->dNb, (size_t)desc->nq * nb * 4, &d.dNb))) return rc;
    if ((rc = dev_copy(fe, desc->N1, (size_t)desc->nq * 4, &d.N1))) return rc;
    if ((rc = dev_copy_transposed(fe, desc->cell_u, nc, 30, &d.cu))) return rc;
    if ((rc = dev_copy_transposed(fe, desc->cell_p, nc, 4, &d.cp))) return rc;
    if ((rc = dev_copy_transposed(fe, desc->cell_b, nc, nb, &d.cb))) return rc;
    const double zero = 0.0;
    if ((rc = dev_copy(fe, desc->n_u_diri ? desc->u_diri : &zero, (size_t)std::max<int64_t>(1, desc->n_u_diri),
                       &d.u_diri)))
        return rc;
    if ((rc = dev_copy(fe, desc->n_b_diri ? desc->b_diri : &zero, (size_t)std::max<int64_t>(1, desc->n_b_diri),
                       &d.b_diri)))
        return rc;
    // inverted index: destination row -> transposed local slots (i * ncell + cell), cell-ascending
    std::vector<int64_t> gptr((size_t)desc->n_b + 1, 0);
    for (int64_t k = 0; k < nc * nb; ++k)
        if (desc->cell_b[k] >= 0) ++gptr[(size_t)desc->cell_b[k] + 1];
    for (int64_t r = 0; r < desc->n_b; ++r) gptr[r + 1] += gptr[r];
    std::vector<int32_t> gidx((size_t)gptr[desc->n_b]);
    std::vector<int64_t> next(gptr.begin(), gptr.end() - 1);
    for (int64_t c = 0; c < nc; ++c)
        for (int i = 0; i < nb; ++i) {
            const int32_t r = desc->cell_b[(size_t)c * nb + i];
            if (r >= 0) gidx[(size_t)next[r]++] = (int32_t)((int64_t)i * nc + c);
        }
    const int64_t *gp;
    const int32_t *gi;
    if ((rc = dev_copy(fe, gptr.data(), gptr.size(), &gp))) return rc;
    if ((rc = dev_copy(fe, gidx.data(), gidx.size(), &gi))) return rc;
    fe->gptr = const_cast<int64_t *>(gp);
    fe->gidx = const_cast<int32_t *>(gi);
    {
        // inversion rows: row -> (l * ncell + cell), l = 3 i + a (velocity node i, component a) or 30 + m (pressure vertex m)
        NPG_REQUIRE(34 * nc < INT32_MAX, "npg_fe_create: too many cells for the 32-bit inverted index");
        std::vector<int64_t> ip((size_t)desc->n_inv + 1, 0);
        for (int64_t k = 0; k < nc * 30; ++k)
            if (desc->cell_u[k] >= 0) ++ip[(size_t)desc->cell_u[k] + 1];
        for (int64_t k = 0; k < nc * 4; ++k)
            if (desc->cell_p[k] >= 0) ++ip[(size_t)desc->cell_p[k] + 1];
        for (int64_t r = 0; r < desc->n_inv; ++r) ip[r + 1] += ip[r];
        std::vector<int32_t> ii((size_t)ip[desc->n_inv]);
        std::vector<int64_t> nx(ip.begin(), ip.end() - 1);
        for (int64_t c = 0; c < nc; ++c) {
            for (int l = 0; l < 30; ++l) {
                const int32_t r = desc->cell_u[(size_t)c * 30 + l];
                if (r >= 0) ii[(size_t)nx[r]++] = (int32_t)((int64_t)l * nc + c);
            }
            for (int m = 0; m < 4; ++m) {
                const int32_t r = desc->cell_p[(size_t)c * 4 + m];
                if (r >= 0) ii[(size_t)nx[r]++] = (int32_t)((int64_t)(30 + m) * nc + c);
            }
        }
        const int64_t *ipd;
        const int32_t *iid;
        if ((rc = dev_copy(fe, ip.data(), ip.size(), &ipd))) return rc;
        if ((rc = dev_copy(fe, ii.data(), ii.size(), &iid))) return rc;
        fe->iptr = const_cast<int64_t *>(ipd);
        fe->iidx = const_cast<int32_t *>(iid);
    }
    NPG_HIP(hipMalloc((void **)&fe->loc, (size_t)nc * nb * sizeof(double)));
    fe->allocs.push_back(fe->loc);
    NPG_HIP(hipMalloc((void **)&fe->missing, sizeof(int)));
    fe->allocs.push_back(fe->missing);
    NPG_HIP(hipMemset(fe->missing, 0, sizeof(int)));
    *out = fe;
    return NPG_OK;
}

NPG_API int npg_fe_destroy(npg_fe *fe) {
    if (!fe) return NPG_OK;
    hipStreamSynchronize(fe->ctx->stream);
    for (void *p : fe->allocs) hipFree(p);
    delete fe;
    return NPG_OK;
}

NPG_API int npg_fe_set_precision(npg_fe *fe, int precision) {
    NPG_REQUIRE(fe, "npg_fe_set_precision: NULL handle");
    NPG_REQUIRE(precision == NPG_FE_FP64 || precision == NPG_FE_FP32,
                "npg_fe_set_precision: precision must be NPG_FE_FP64 or NPG_FE_FP32");
    fe->precision = precision;
    return NPG_OK;
}

NPG_API int npg_fe_get_precision(const npg_fe *fe) { return fe ? fe->precision : NPG_EINVAL; }

static int coef_index(const char *name) {
    if (!strcmp(name, "nu")) return 0;
    if (!strcmp(name, "kappa_h")) return 1;
    if (!strcmp(name, "kappa_v")) return 2;
    if (!strcmp(name, "f")) return 3;
    return -1;
}

static void refresh_coef_ptrs(npg_fe *fe) {
    fe->d.nu = fe->coef[0];
    fe->d.kh = fe->coef[1];
    fe->d.kv = fe->coef[2];
    fe->d.f = fe->coef[3];
}

static int ensure_coef(npg_fe *fe, int k) {
    if (fe->coef[k]) return NPG_OK;
    NPG_HIP(hipMalloc((void **)&fe->coef[k], (size_t)fe->d.ncell * fe->d.nq * sizeof(double)));
    fe->allocs.push_back(fe->coef[k]);
    refresh_coef_ptrs(fe);
    return NPG_OK;
}

NPG_API int npg_fe_set_coeff(npg_fe *fe, const char *name, const double *values) {
    NPG_REQUIRE(fe && name && values, "npg_fe_set_coeff: NULL argument");
    const int k = coef_index(name);
    NPG_REQUIRE(k >= 0, "npg_fe_set_coeff: unknown coefficient '%s'", name);
    int rc = ensure_coef(fe, k);
    if (rc) return rc;
    const int64_t nc = fe->d.ncell;
    const int nq = fe->d.nq;
    std::vector<double> tmp((size_t)nc * nq);
    for (int64_t c = 0; c < nc; ++c)
        for (int q = 0; q < nq; ++q) tmp[(size_t)q * nc + c] = values[(size_t)c * nq + q];
    // kernels that read the table run on ctx->stream (non-blocking: not ordered with the null stream) - drain it first
    NPG_HIP(hipStreamSynchronize(fe->ctx->stream));
    NPG_HIP(hipMemcpy(fe->coef[k], tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    if (k == 2) {      // remember the background kappa_v for the convection closure
        if (!fe->kv0) {
            NPG_HIP(hipMalloc((void **)&fe->kv0, tmp.size() * sizeof(double)));
            fe->allocs.push_back(fe->kv0);
        }
        NPG_HIP(hipMemcpy(fe->kv0, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return NPG_OK;
}

static inline int cell_grid(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

static int check_state_vectors(const npg_fe *fe, const npg_vec *b, const npg_vec *bp, const npg_vec *xi,
                               const npg_vec *xip) {
    NPG_REQUIRE(b && bp && xi && xip, "fe: NULL state vector");
    NPG_REQUIRE(b->n == fe->n_b && bp->n == fe->n_b, "fe: buoyancy vectors must have %lld entries", (long long)fe->n_b);
    NPG_REQUIRE(xi->n == fe->n_inv && xip->n == fe->n_inv, "fe: inversion vectors must have %lld entries",
                (long long)fe->n_inv);
    return NPG_OK;
}

static int advection_pass1(npg_fe *fe, int scheme, double dt, double N2, const npg_vec *b, const npg_vec *bp,
                           const npg_vec *xi, const npg_vec *xip) {
    NPG_REQUIRE(scheme == NPG_BDF1 || scheme == NPG_BDF2, "fe: scheme must be NPG_BDF1 or NPG_BDF2");
    int rc = check_state_vectors(fe, b, bp, xi, xip);
    if (rc) return rc;
    const int grid = cell_grid(fe->d.ncell);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, fe->ctx->stream, fe->d, scheme, dt, N2, b->d, bp->d, xi->d,
                           xip->d, fe->loc);
    };
    const bool f32 = fe->precision == NPG_FE_FP32;
    if (fe->d.nb == 10) f32 ? go(k_advection_local<float, 10>) : go(k_advection_local<double, 10>);
    else f32 ? go(k_advection_local<float, 4>) : go(k_advection_local<double, 4>);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

static int gather_pass2(npg_fe *fe, const RhsTerms &rt, npg_vec *out) {
    NPG_REQUIRE(out && out->n == fe->n_b, "fe: output vector must have %lld entries", (long long)fe->n_b);
    const int grid = std::min(2048, cell_grid(fe->n_b));
    hipLaunchKernelGGL(k_gather_rows, dim3(grid), dim3(kBlock), 0, fe->ctx->stream, fe->gptr, fe->gidx, fe->loc,
                       fe->n_b, rt, out->d);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

NPG_API int npg_fe_advection_rhs(npg_fe *fe, int scheme, double dt, double N2, const npg_vec *b, const npg_vec *b_prev,
                                 const npg_vec *x_inv, const npg_vec *x_inv_prev, npg_vec *out) {
    NPG_REQUIRE(fe, "npg_fe_advection_rhs: NULL handle");
    int rc = advection_pass1(fe, scheme, dt, N2, b, b_prev, x_inv, x_inv_prev);
    if (rc) return rc;
    RhsTerms rt{};
    return gather_pass2(fe, rt, out);
}

NPG_API int npg_fe_evolution_rhs(npg_fe *fe, int scheme, double dt, double N2, double theta, const npg_vec *b,
                                 const npg_vec *b_prev, const npg_vec *x_inv, const npg_vec *x_inv_prev,
                                 const npg_vec *rhs_diff, const npg_vec *rhs_flux, const npg_vec *rhs_M,
                                 const npg_vec *rhs_h, const npg_vec *rhs_v, npg_vec *y) {
    NPG_REQUIRE(fe, "npg_fe_evolution_rhs: NULL handle");
    const npg_vec *opt[5] = {rhs_diff, rhs_flux, rhs_M, rhs_h, rhs_v};
    for (const npg_vec *v : opt)
        NPG_REQUIRE(!v || v->n == fe->n_b, "npg_fe_evolution_rhs: rhs_* vectors must have %lld entries",
                    (long long)fe->n_b);
    int rc = advection_pass1(fe, scheme, dt, N2, b, b_prev, x_inv, x_inv_prev);
    if (rc) return rc;
    RhsTerms rt{};
    rt.theta = theta;
    rt.dt = dt;
    rt.rhs_diff = rhs_diff ? rhs_diff->d : nullptr;
    rt.rhs_flux = rhs_flux ? rhs_flux->d : nullptr;
    rt.rhs_M = rhs_M ? rhs_M->d : nullptr;
    rt.rhs_h = rhs_h ? rhs_h->d : nullptr;
    rt.rhs_v = rhs_v ? rhs_v->d : nullptr;
    return gather_pass2(fe, rt, y);
}

NPG_API int npg_fe_assemble_rhs_diff(npg_fe *fe, double N2, npg_vec *out) {
    NPG_REQUIRE(fe && out, "npg_fe_assemble_rhs_diff: NULL argument");
    NPG_REQUIRE(fe->d.kv, "npg_fe_assemble_rhs_diff: coefficient kappa_v has not been set");
    const int grid = cell_grid(fe->d.ncell);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, fe->ctx->stream, fe->d, N2, fe->loc);
    };
    const bool f32 = fe->precision == NPG_FE_FP32;
    if (fe->d.nb == 10) f32 ? go(k_rhs_diff_local<float, 10>) : go(k_rhs_diff_local<double, 10>);
    else f32 ? go(k_rhs_diff_local<float, 4>) : go(k_rhs_diff_local<double, 4>);
    NPG_HIP(hipGetLastError());
    RhsTerms rt{};
    return gather_pass2(fe, rt, out);
}

NPG_API int npg_fe_assemble_matrix(npg_fe *fe, int which, double scale, int full_stress, npg_csr *A, npg_vec *lift) {
    NPG_REQUIRE(fe && A, "npg_fe_assemble_matrix: NULL argument");
    NPG_REQUIRE(A->nnode() == 0, "npg_fe_assemble_matrix: the matrix is stored by node blocks and cannot be re-assembled");
    NPG_REQUIRE(A->lb_nblocks == 0, "npg_fe_assemble_matrix: the matrix holds dense line-block packs (npg_csr_line_block_inverse)");
    hipStream_t st = fe->ctx->stream;
    const FeDev &d = fe->d;
    const bool f32 = fe->precision == NPG_FE_FP32;
    NPG_HIP(hipMemsetAsync(A->val, 0, (size_t)A->nnz * sizeof(double), st));
    NPG_HIP(hipMemsetAsync(fe->missing, 0, sizeof(int), st));
    if (lift) NPG_HIP(hipMemsetAsync(lift->d, 0, (size_t)lift->n * sizeof(double), st));
    switch (which) {
        case NPG_MAT_M:
        case NPG_MAT_KH:
        case NPG_MAT_KV: {
            // (a mesh-partitioned rank assembles its OWNED rows only: they lead the local numbering, and their columns -
            // owned + first ghost layer - lead the column numbering; the engine's vectors carry further ghosts behind them)
            NPG_REQUIRE(A->m <= fe->n_b && A->n <= fe->n_b && A->m <= A->n,
                        "npg_fe_assemble_matrix: matrix must be n_b x n_b (or the leading owned rows x leading columns of it)");
            NPG_REQUIRE(!lift || (lift->n >= A->m && lift->n <= fe->n_b), "npg_fe_assemble_matrix: lift must have n_b entries");
            NPG_REQUIRE(which == NPG_MAT_M || (which == NPG_MAT_KH ? d.kh : d.kv),
                        "npg_fe_assemble_matrix: diffusivity coefficient has not been set");
            const int grid = cell_grid(A->m * kQL);
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, st, d, which, fe->gptr, fe->gidx, A->m, A->rowptr,
                                   A->col, A->val, lift ? lift->d : nullptr, fe->missing);
            };
            if (d.nb == 10) f32 ? go(k_assemble_b<float, 10>) : go(k_assemble_b<double, 10>);
            else f32 ? go(k_assemble_b<float, 4>) : go(k_assemble_b<double, 4>);
            break;
        }
        case NPG_MAT_A: {
            NPG_REQUIRE(A->m <= fe->n_inv && A->n <= fe->n_inv && A->m <= A->n,
                        "npg_fe_assemble_matrix: A must be n_inv x n_inv (or the leading owned rows x leading columns of it)");
            NPG_REQUIRE(d.nu && d.f, "npg_fe_assemble_matrix: coefficients nu and f must be set");
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(cell_grid(A->m * kQL)), dim3(kBlock), 0, st, d, scale, full_stress,
                                   fe->iptr, fe->iidx, A->m, A->rowptr, A->col, A->val, fe->missing);
            };
            f32 ? go(k_assemble_A<float>) : go(k_assemble_A<double>);
            break;
        }
        case NPG_MAT_B: {
            NPG_REQUIRE(A->m <= fe->n_inv && A->n == fe->n_b, "npg_fe_assemble_matrix: B must be n_inv (or its leading owned rows) x n_b");
            NPG_REQUIRE(!lift || (lift->n >= A->m && lift->n <= fe->n_inv), "npg_fe_assemble_matrix: lift must have n_inv entries");
            const int grid = cell_grid(A->m * kQL);
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, st, d, scale, fe->iptr, fe->iidx, A->m, A->rowptr,
                                   A->col, A->val, lift ? lift->d : nullptr, fe->missing);
            };
            if (d.nb == 10) f32 ? go(k_assemble_B<float, 10>) : go(k_assemble_B<double, 10>);
            else f32 ? go(k_assemble_B<float, 4>) : go(k_assemble_B<double, 4>);
            break;
        }
        default:
            NPG_REQUIRE(false, "npg_fe_assemble_matrix: unknown matrix id %d", which);
    }
    NPG_HIP(hipGetLastError());
    int miss = 0;
    NPG_HIP(hipMemcpyAsync(&miss, fe->missing, sizeof(int), hipMemcpyDeviceToHost, st));
    NPG_HIP(hipStreamSynchronize(st));
    NPG_REQUIRE(miss == 0, "npg_fe_assemble_matrix: %d non-zero local entries fall outside the CSR pattern", miss);
    return csr_repack(A);           // (a record-form companion follows the assembled values: npg_csr_pack_nodes)
}

static int coeff_update(npg_fe *fe, int mode, const npg_vec *b, double p0, double p1, double alpha, double N2,
                        double p2, double p3, const double *base, double *out) {
    const int grid = cell_grid(fe->d.ncell);
    if (fe->d.nb == 10)
        hipLaunchKernelGGL(k_coeff_from_bz<10>, dim3(grid), dim3(kBlock), 0, fe->ctx->stream, fe->d, mode, b->d, base,
                           p0, p1, alpha, N2, p2, p3, out);
    else
        hipLaunchKernelGGL(k_coeff_from_bz<4>, dim3(grid), dim3(kBlock), 0, fe->ctx->stream, fe->d, mode, b->d, base, p0,
                           p1, alpha, N2, p2, p3, out);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

NPG_API int npg_fe_update_kappa_convection(npg_fe *fe, const double *kappa_v0_host_or_null, double kappa_c,
                                           double N2min, double alpha, double N2, const npg_vec *b) {
    NPG_REQUIRE(fe && b && b->n == fe->n_b, "npg_fe_update_kappa_convection: bad argument");
    if (kappa_v0_host_or_null) {
        int rc = npg_fe_set_coeff(fe, "kappa_v", kappa_v0_host_or_null);
        if (rc) return rc;
    }
    NPG_REQUIRE(fe->kv0 && fe->coef[2], "npg_fe_update_kappa_convection: background kappa_v has not been set");
    return coeff_update(fe, 0, b, kappa_c, N2min, alpha, N2, 0.0, 0.0, fe->kv0, fe->coef[2]);
}

NPG_API int npg_fe_update_nu_eddy(npg_fe *fe, double N2min, double alpha, double N2, double smoothing, double nu_min,
                                  const npg_vec *b) {
    NPG_REQUIRE(fe && b && b->n == fe->n_b, "npg_fe_update_nu_eddy: bad argument");
    NPG_REQUIRE(fe->coef[3], "npg_fe_update_nu_eddy: coefficient f has not been set");
    int rc = ensure_coef(fe, 0);
    if (rc) return rc;
    return coeff_update(fe, 1, b, 0.0, N2min, alpha, N2, smoothing, nu_min, nullptr, fe->coef[0]);
}

// coarse[q][c] (every q) = sum_k |K_k| mean_q(fine[.][8c + k]) / sum_k |K_k|  over the eight children 8c .. 8c + 7 of coarse cell c
__global__ void __launch_bounds__(kBlock) k_coeff_restrict(const double *__restrict__ fine, const double *__restrict__ wdet, const double *__restrict__ qw,
                                                           int nq, int64_t ncf, int64_t ncc, int nqc, double *__restrict__ coarse) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= ncc) return;
    double sw = 0.0;
    for (int q = 0; q < nq; ++q) sw += qw[q];
    double num = 0.0, den = 0.0;
    for (int k = 0; k < 8; ++k) {
        const int64_t cf = 8 * c + k;
        double m = 0.0;
        for (int q = 0; q < nq; ++q) m += qw[q] * fine[(size_t)q * ncf + cf];
        num += wdet[cf] * (m / sw);
        den += wdet[cf];
    }
    const double v = num / den;
    for (int q = 0; q < nqc; ++q) coarse[(size_t)q * ncc + c] = v;
}

// out[c] = quadrature mean of a coefficient table over cell c
__global__ void __launch_bounds__(kBlock) k_coeff_cell_mean(const double *__restrict__ tab, const double *__restrict__ qw, int nq, int64_t nc,
                                                            double *__restrict__ out) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= nc) return;
    double sw = 0.0, m = 0.0;
    for (int q = 0; q < nq; ++q) {
        sw += qw[q];
        m += qw[q] * tab[(size_t)q * nc + c];
    }
    out[c] = m / sw;
}

NPG_API int npg_fe_coeff_cell_mean(const npg_fe *fe, const char *name, npg_vec *out) {
    NPG_REQUIRE(fe && name && out, "npg_fe_coeff_cell_mean: NULL argument");
    const int k = coef_index(name);
    NPG_REQUIRE(k >= 0, "npg_fe_coeff_cell_mean: unknown coefficient '%s'", name);
    NPG_REQUIRE(fe->coef[k], "npg_fe_coeff_cell_mean: coefficient '%s' has not been set", name);
    NPG_REQUIRE(out->n == fe->d.ncell && out->ctx == fe->ctx, "npg_fe_coeff_cell_mean: the output vector must hold one entry per cell (%lld)",
                (long long)fe->d.ncell);
    hipLaunchKernelGGL(k_coeff_cell_mean, dim3(cell_grid(fe->d.ncell)), dim3(kBlock), 0, fe->ctx->stream, (const double *)fe->coef[k], fe->d.qw, fe->d.nq,
                       fe->d.ncell, out->d);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

NPG_API int npg_fe_restrict_coeff(npg_fe *coarse, const npg_fe *fine, const char *name) {
    NPG_REQUIRE(coarse && fine && name, "npg_fe_restrict_coeff: NULL argument");
    const int k = coef_index(name);
    NPG_REQUIRE(k >= 0, "npg_fe_restrict_coeff: unknown coefficient '%s'", name);
    NPG_REQUIRE(coarse->ctx == fine->ctx, "npg_fe_restrict_coeff: the two element engines live on different contexts");
    NPG_REQUIRE(fine->d.ncell == 8 * coarse->d.ncell, "npg_fe_restrict_coeff: the fine mesh (%lld cells) is not the uniform refinement of the coarse one (%lld)",
                (long long)fine->d.ncell, (long long)coarse->d.ncell);
    NPG_REQUIRE(fine->coef[k], "npg_fe_restrict_coeff: coefficient '%s' has not been set on the fine mesh", name);
    int rc = ensure_coef(coarse, k);
    if (rc) return rc;
    hipLaunchKernelGGL(k_coeff_restrict, dim3(cell_grid(coarse->d.ncell)), dim3(kBlock), 0, coarse->ctx->stream, (const double *)fine->coef[k], fine->d.wdet,
                       fine->d.qw, fine->d.nq, fine->d.ncell, coarse->d.ncell, coarse->d.nq, coarse->coef[k]);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

NPG_API int npg_fe_cfl_ratio(npg_fe *fe, const double *h_cells_host, double u_min, const npg_vec *x_inv, double *out) {
    NPG_REQUIRE(fe && x_inv && out && x_inv->n == fe->n_inv, "npg_fe_cfl_ratio: bad argument");
    if (h_cells_host) {
        if (!fe->hcell) {
            NPG_HIP(hipMalloc((void **)&fe->hcell, (size_t)fe->d.ncell * sizeof(double)));
            fe->allocs.push_back(fe->hcell);
        }
        NPG_HIP(hipStreamSynchronize(fe->ctx->stream));
        NPG_HIP(hipMemcpy(fe->hcell, h_cells_host, (size_t)fe->d.ncell * sizeof(double), hipMemcpyHostToDevice));
    }
    NPG_REQUIRE(fe->hcell, "npg_fe_cfl_ratio: cell sizes have not been provided");
    npg_ctx *ctx = fe->ctx;
    const int grid = std::min(1024, cell_grid(fe->d.ncell));
    hipLaunchKernelGGL(k_cfl, dim3(grid), dim3(kBlock), 0, ctx->stream, fe->d, fe->hcell, u_min, x_inv->d,
                       ctx->d_scratch + 8);
    NPG_HIP(hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch + 8, grid * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    double m = 1e300;
    for (int i = 0; i < grid; ++i) m = std::fmin(m, ctx->h_scratch[i]);
    *out = m;
    return NPG_OK;
}
